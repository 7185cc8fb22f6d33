// NOT COMPILED INTO THE LIBRARY.  wt_qk_rows_batch as fp32 MFMA tiles (round 2, measured 118 us per 32 whisper-base
// windows against 108 us for the thread-per-frame kernel in csrc/wt_capture.hip; correct against the torch reference:
// tests/test_gpu_parity.py::test_qk_rows_batch_vs_torch_and_single_window and the batched end-to-end cases passed with
// it).  Kept for the record: it replaces the kernel of the same name (launch grid: ((n_ctx + 127) / 128, n_sel,
// n_batch), template arguments <ST, DT>).  See DESIGN.md section 6.
//
// One 32 x 32 x 2 fp32 MFMA step takes, per lane (j = lane % 32, p = lane / 32), A[j][p] and B[p][j]: with the head
// dimension split by parity -- step m covers dimensions 2m and 2m + 1 -- lane (j, p) always needs dimension 2m + p of
// query row j and of frame j.  Tiles are therefore kept in LDS as [parity][row][dimension / 2] with a row pitch of 36
// floats: a lane's 32 values are 8 ds_read_b128, conflict-free for the 16-lane groups of that instruction.
constexpr int QT_ROWS = 32;                 // rows of a tile (query rows / frames)
constexpr int QT_PITCH = 36;                // floats per row of a parity plane (32 used)
constexpr int QT_PLANE = QT_ROWS * QT_PITCH + 16;   // second plane 16 banks off the first (staging writes alternate planes)
struct QkTile {
    float v[2 * QT_PLANE];
    __device__ __forceinline__ float &at(int row, int d) { return v[(d & 1) * QT_PLANE + row * QT_PITCH + (d >> 1)]; }
    __device__ __forceinline__ const float4 *fragment(int row, int parity) const {
        return reinterpret_cast<const float4 *>(v + parity * QT_PLANE + row * QT_PITCH);
    }
};

typedef float qk_acc_t __attribute__((ext_vector_type(16)));

template <typename T, typename DT>
__global__ __launch_bounds__(256) void qk_rows_batch_kernel(QkLayers L, int n_q, int64_t q_bstride, int64_t k_bstride, int n_ctx,
                                                            int d_model, float scale, const int32_t *__restrict__ sel_layer,
                                                            const int32_t *__restrict__ sel_head,
                                                            const int32_t *__restrict__ sel_slot,
                                                            const int32_t *__restrict__ row_begin,
                                                            const int32_t *__restrict__ row_end, DT *__restrict__ ring,
                                                            int64_t ring_bstride, int64_t ring_rows, int64_t ring_row0) {
    constexpr int HD = 64;
    __shared__ __attribute__((aligned(16))) QkTile qt;        // the workgroup's current 32 query rows
    __shared__ __attribute__((aligned(16))) QkTile kt[4];     // one 32-frame tile per wave
    const int s = blockIdx.y, b = blockIdx.z;
    const int layer = sel_layer[s], h = sel_head[s];
    const int r_lo = row_begin ? max(row_begin[b], 0) : 0;
    const int r_hi = row_end ? min(row_end[b], n_q) : n_q;
    if (r_lo >= r_hi) return;   // block-uniform
    const T *q = static_cast<const T *>(L.q[layer]) + (int64_t)b * q_bstride + (int64_t)h * HD;
    const T *k = static_cast<const T *>(L.k[layer]) + (int64_t)b * k_bstride + (int64_t)h * HD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, parity = lane >> 5;
    const int f0 = blockIdx.x * (4 * QT_ROWS) + wave * QT_ROWS;   // first frame of this wave's tile
    const int f = f0 + j;                                        // the frame this lane holds (both halves of the wave)
    // K tile of the wave: 32 frames x 64 dimensions, scaled once, read coalesced (a wave instruction = one 256-byte row)
#pragma unroll 8
    for (int e = lane; e < QT_ROWS * HD; e += 64) {
        const int row = e >> 6, d = e & 63;
        kt[wave].at(row, d) = scaled<T>(k[(int64_t)min(f0 + row, n_ctx - 1) * d_model + d], scale);
    }
    float kb[32];   // (same-wave LDS order: the wave reads back what it wrote)
    {
        const float4 *kf = kt[wave].fragment(j, parity);
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4) {
            const float4 x = kf[m4];
            kb[4 * m4] = x.x; kb[4 * m4 + 1] = x.y; kb[4 * m4 + 2] = x.z; kb[4 * m4 + 3] = x.w;
        }
    }
    DT *out = ring + (int64_t)b * ring_bstride + ((int64_t)sel_slot[s] * ring_rows + ring_row0) * n_ctx + f;
    for (int r0 = r_lo; r0 < r_hi; r0 += QT_ROWS) {
        __syncthreads();   // the previous trip's readers are done
        for (int e = threadIdx.x; e < QT_ROWS * HD; e += 256) {
            const int row = e >> 6, d = e & 63;
            qt.at(row, d) = r0 + row < r_hi ? scaled<T>(q[(int64_t)(r0 + row) * d_model + d], scale) : 0.f;
        }
        __syncthreads();
        qk_acc_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float4 *qf = qt.fragment(j, parity);
#pragma unroll
        for (int m4 = 0; m4 < 8; ++m4) {
            const float4 a = qf[m4];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kb[4 * m4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kb[4 * m4 + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kb[4 * m4 + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kb[4 * m4 + 3], acc, 0, 0, 0);
        }
        // D[i][j]: register v of lane (j, p) holds row i = 8 (v / 4) + 4 p + v % 4, column j
        if (f < n_ctx) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int r = r0 + 8 * (v >> 2) + 4 * parity + (v & 3);
                float x = acc[v];
                if (sizeof(T) == 2) x = __half2float(__float2half(x));
                if (r < r_hi) out[(int64_t)r * n_ctx] = cvt<float, DT>(x);
            }
        }
    }
}
