// Device-side capture of the cross-attention rows the alignment needs.
//
// Replaces /root/reference/whisper_timestamped/transcribe.py:783-793
// (hook_attention_weights): the reference keeps, for every decoded token and
// every hooked decoder layer, the LAST query row of the layer's QK logits
// (1, H, n_q, 1500) -- after a device->host copy (`w.cpu()`, :793) of all H
// heads.  Here the rows of the SELECTED alignment heads only are copied into a
// preallocated device ring (A_sel, ring_rows, n_ctx), one launch per layer,
// no synchronisation, no host traffic.  Pure HBM copy: n_sel * n_ctx * 4 bytes
// read + written per call.
#include <hip/hip_fp16.h>

#include "wt_common.h"

namespace wt {

template <typename ST, typename DT>
__device__ __forceinline__ DT cvt(ST v);
template <> __device__ __forceinline__ float cvt<float, float>(float v) { return v; }
template <> __device__ __forceinline__ __half cvt<float, __half>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ float cvt<__half, float>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ __half cvt<__half, __half>(__half v) { return v; }

template <typename ST, typename DT>
__global__ __launch_bounds__(256) void capture_rows_kernel(const ST *__restrict__ qk, int n_q, int n_ctx,
                                                           const int32_t *__restrict__ heads, const int32_t *__restrict__ slots,
                                                           DT *__restrict__ ring, int64_t ring_rows, int64_t row) {
    const int s = blockIdx.y;
    const ST *src = qk + ((int64_t)heads[s] * n_q + (n_q - 1)) * n_ctx;  // last query row of that head
    DT *dst = ring + ((int64_t)slots[s] * ring_rows + row) * n_ctx;
    for (int f = blockIdx.x * 256 + threadIdx.x; f < n_ctx; f += gridDim.x * 256) dst[f] = cvt<ST, DT>(src[f]);
}

int capture_rows(const void *qk, int qk_dtype, int n_heads, int n_q, int n_ctx, const int32_t *heads, const int32_t *slots,
                 int n_sel, void *ring, int ring_dtype, int64_t ring_rows, int64_t row, hipStream_t st) {
    if (!qk || !heads || !slots || !ring || n_heads <= 0 || n_q <= 0 || n_ctx <= 0 || n_sel < 0 || row < 0 || row >= ring_rows) {
        set_error("wt_capture_rows: bad argument (row=%lld of %lld)", (long long)row, (long long)ring_rows);
        return WT_E_BADARG;
    }
    if (n_sel == 0) return WT_OK;
    const dim3 grid((n_ctx + 255) / 256 > 8 ? 8 : (n_ctx + 255) / 256, n_sel), block(256);
#define WT_CAP(ST, DT)                                                                                            \
    hipLaunchKernelGGL((capture_rows_kernel<ST, DT>), grid, block, 0, st, (const ST *)qk, n_q, n_ctx, heads, slots, \
                       (DT *)ring, ring_rows, row)
    if (qk_dtype == WT_DTYPE_F32 && ring_dtype == WT_DTYPE_F32) WT_CAP(float, float);
    else if (qk_dtype == WT_DTYPE_F32 && ring_dtype == WT_DTYPE_F16) WT_CAP(float, __half);
    else if (qk_dtype == WT_DTYPE_F16 && ring_dtype == WT_DTYPE_F32) WT_CAP(__half, float);
    else if (qk_dtype == WT_DTYPE_F16 && ring_dtype == WT_DTYPE_F16) WT_CAP(__half, __half);
    else {
        set_error("wt_capture_rows: dtype %d -> %d", qk_dtype, ring_dtype);
        return WT_E_BADARG;
    }
#undef WT_CAP
    WT_HIP(hipGetLastError());
    return WT_OK;
}


// ---------------------------------------------------------------------------
// QK rows of the alignment heads computed from the projections themselves.
//
// The reference can only observe qk as the second output of whisper's MultiHeadAttention, which exists only on the
// slow, unfused attention path (it wraps every decode in whisper.model.disable_sdpa(), transcribe.py:49-61,903):
// ALL attention modules, encoder included, then materialise their (H, n_q, n_k) score matrices.  The word alignment
// needs A_sel heads of the decoder's cross-attention, one query row per token.  This kernel computes exactly those
// rows from q = cross_attn.query(x) and K = cross_attn.key(xa) (both observable with forward hooks):
//     qk[h, r, f] = sum_d (q[r, h*hd + d] * scale) * (K[f, h*hd + d] * scale),   scale = hd ** -0.25
// (whisper/model.py qkv_attention), so the model itself can stay on the fused attention path.
// For an fp16 model the scaled operands and the result are rounded to fp16 like the backend's own matmul does.
template <typename T>
__device__ __forceinline__ float scaled(T v, float scale);
template <> __device__ __forceinline__ float scaled<float>(float v, float scale) { return v * scale; }
template <> __device__ __forceinline__ float scaled<__half>(__half v, float scale) {
    return __half2float(__float2half(__half2float(v) * scale));
}

template <typename T, typename DT>
__global__ __launch_bounds__(256) void qk_rows_kernel(const T *__restrict__ q, const T *__restrict__ k, int n_ctx, int d_model,
                                                      int head_dim, float scale, const int32_t *__restrict__ heads,
                                                      const int32_t *__restrict__ slots, DT *__restrict__ ring,
                                                      int64_t ring_rows, int64_t row0) {
    __shared__ float qs[256];                       // this head's scaled query row (head_dim <= 256)
    const int s = blockIdx.y, r = blockIdx.z;
    const int h = heads[s];
    const T *qrow = q + (int64_t)r * d_model + (int64_t)h * head_dim;
    if ((int)threadIdx.x < head_dim) qs[threadIdx.x] = scaled<T>(qrow[threadIdx.x], scale);
    __syncthreads();
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n_ctx) return;
    const T *krow = k + (int64_t)f * d_model + (int64_t)h * head_dim;
    float acc = 0.f;
    for (int d = 0; d < head_dim; ++d) acc = fmaf(qs[d], scaled<T>(krow[d], scale), acc);
    if (sizeof(T) == 2) acc = __half2float(__float2half(acc));
    ring[((int64_t)slots[s] * ring_rows + row0 + r) * n_ctx + f] = cvt<float, DT>(acc);
}

int qk_rows(const void *q, const void *k, int dtype, int n_rows, int n_ctx, int d_model, int head_dim, float scale,
            const int32_t *heads, const int32_t *slots, int n_sel, void *ring, int ring_dtype, int64_t ring_rows, int64_t row0,
            hipStream_t st) {
    if (!q || !k || !heads || !slots || !ring || n_rows <= 0 || n_ctx <= 0 || d_model <= 0 || head_dim <= 0 || head_dim > 256 ||
        d_model % head_dim != 0 || n_sel < 0 || row0 < 0 || row0 + n_rows > ring_rows) {
        set_error("wt_qk_rows: bad argument (rows %lld..%lld of %lld, head_dim %d)", (long long)row0, (long long)(row0 + n_rows),
                  (long long)ring_rows, head_dim);
        return WT_E_BADARG;
    }
    if (n_sel == 0) return WT_OK;
    const dim3 grid((n_ctx + 255) / 256, n_sel, n_rows), block(256);
#define WT_QK(ST, DT)                                                                                                 \
    hipLaunchKernelGGL((qk_rows_kernel<ST, DT>), grid, block, 0, st, (const ST *)q, (const ST *)k, n_ctx, d_model, head_dim, \
                       scale, heads, slots, (DT *)ring, ring_rows, row0)
    if (dtype == WT_DTYPE_F32 && ring_dtype == WT_DTYPE_F32) WT_QK(float, float);
    else if (dtype == WT_DTYPE_F32 && ring_dtype == WT_DTYPE_F16) WT_QK(float, __half);
    else if (dtype == WT_DTYPE_F16 && ring_dtype == WT_DTYPE_F32) WT_QK(__half, float);
    else if (dtype == WT_DTYPE_F16 && ring_dtype == WT_DTYPE_F16) WT_QK(__half, __half);
    else {
        set_error("wt_qk_rows: dtype %d -> %d", dtype, ring_dtype);
        return WT_E_BADARG;
    }
#undef WT_QK
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// ---------------------------------------------------------------------------
// The same rows for a BATCH of windows and ALL hooked layers in one launch (the batched naive strategy: B
// independent 30 s windows teacher-forced through the decoder at once).  Per (window b, selected head s) the work
// is a small GEMM  Q_b[rows, hd] x K_b[n_ctx, hd]^T: one thread owns one frame f, keeps K_b[f, h*hd .. +hd) -- scaled
// once -- in registers and walks the query rows, which the workgroup stages (scaled) in LDS and reads back with
// broadcast ds_read_b128: K crosses L2 once per (window, head) instead of once per query row, the stores of a row
// are coalesced over f.  Same arithmetic as qk_rows_kernel (sequential fma over the head dimension), so the rows
// are bit-identical to the single-window entry.  Rows outside [row_begin[b], row_end[b]) -- prompt rows nobody
// aligns, padding of shorter transcripts -- are skipped.
struct QkLayers {
    const void *q[WT_MAX_LAYERS];
    const void *k[WT_MAX_LAYERS];
};

constexpr int QB_ROWS = 32;   // query rows staged per trip

template <typename T, typename DT, int HD>
__global__ __launch_bounds__(256) void qk_rows_batch_kernel(QkLayers L, int n_q, int64_t q_bstride, int64_t k_bstride, int n_ctx,
                                                            int d_model, float scale, const int32_t *__restrict__ sel_layer,
                                                            const int32_t *__restrict__ sel_head,
                                                            const int32_t *__restrict__ sel_slot,
                                                            const int32_t *__restrict__ row_begin,
                                                            const int32_t *__restrict__ row_end,
                                                            const int32_t *__restrict__ ring_index, DT *__restrict__ ring,
                                                            int64_t ring_bstride, int64_t ring_rows, int64_t ring_row0) {
    __shared__ __attribute__((aligned(16))) float qs[QB_ROWS][HD];
    const int s = blockIdx.y, b = blockIdx.z;
    const int layer = sel_layer[s], h = sel_head[s];
    const int r_lo = row_begin ? max(row_begin[b], 0) : 0;
    const int r_hi = row_end ? min(row_end[b], n_q) : n_q;
    if (r_lo >= r_hi) return;   // block-uniform
    const T *q = static_cast<const T *>(L.q[layer]) + (int64_t)b * q_bstride + (int64_t)h * HD;
    const T *k = static_cast<const T *>(L.k[layer]) + (int64_t)b * k_bstride + (int64_t)h * HD;
    const int f = blockIdx.x * 256 + threadIdx.x;
    const bool f_ok = f < n_ctx;
    float kr[HD];
    {
        const T *krow = k + (int64_t)min(f, n_ctx - 1) * d_model;
#pragma unroll
        for (int d = 0; d < HD; ++d) kr[d] = scaled<T>(krow[d], scale);
    }
    // window b of the batch writes ring block ring_index[b] (streams decoded together need not be neighbours in the ring)
    const int64_t rb = ring_index ? ring_index[b] : b;
    DT *out = ring + rb * ring_bstride + ((int64_t)sel_slot[s] * ring_rows + ring_row0) * n_ctx + f;
    for (int r0 = r_lo; r0 < r_hi; r0 += QB_ROWS) {
        const int nr = min(QB_ROWS, r_hi - r0);
        __syncthreads();   // the previous trip's readers are done
        for (int e = threadIdx.x; e < nr * HD; e += 256) {
            const int r = e / HD, d = e % HD;
            qs[r][d] = scaled<T>(q[(int64_t)(r0 + r) * d_model + d], scale);
        }
        __syncthreads();
        for (int r = 0; r < nr; ++r) {
            // 64 multiply-adds as 32 v_pk_fma_f32 on two independent accumulator pairs (dimensions 4j, 4j+1 | 4j+2,
            // 4j+3): half the VALU instructions of the scalar chain and two chains in flight per lane
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 acc_a = {0.f, 0.f}, acc_b = {0.f, 0.f};
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 qv = *reinterpret_cast<const float4 *>(&qs[r][d]);
                acc_a = __builtin_elementwise_fma((f2){qv.x, qv.y}, (f2){kr[d], kr[d + 1]}, acc_a);
                acc_b = __builtin_elementwise_fma((f2){qv.z, qv.w}, (f2){kr[d + 2], kr[d + 3]}, acc_b);
            }
            float acc = (acc_a.x + acc_a.y) + (acc_b.x + acc_b.y);
            if (sizeof(T) == 2) acc = __half2float(__float2half(acc));
            if (f_ok) out[(int64_t)(r0 + r) * n_ctx] = cvt<float, DT>(acc);
        }
    }
}

int qk_rows_batch(const void *const *q_layers, const void *const *k_layers, int n_layers, int dtype, int n_batch, int n_q,
                  int64_t q_bstride, int64_t k_bstride, int n_ctx, int d_model, int head_dim, float scale,
                  const int32_t *sel_layer, const int32_t *sel_head, const int32_t *sel_slot, int n_sel,
                  const int32_t *row_begin, const int32_t *row_end, const int32_t *ring_index, void *ring, int ring_dtype,
                  int64_t ring_bstride, int64_t ring_rows, int64_t ring_row0, hipStream_t st) {
    if (!q_layers || !k_layers || n_layers <= 0 || n_layers > WT_MAX_LAYERS || !sel_layer || !sel_head || !sel_slot || !ring ||
        n_batch < 0 || n_q <= 0 || n_ctx <= 0 || d_model <= 0 || head_dim <= 0 || d_model % head_dim != 0 || n_sel < 0 ||
        ring_row0 < 0 || ring_row0 + n_q > ring_rows) {
        set_error("wt_qk_rows_batch: bad argument (%d layers, rows %lld..%lld of %lld)", n_layers, (long long)ring_row0,
                  (long long)(ring_row0 + n_q), (long long)ring_rows);
        return WT_E_BADARG;
    }
    if (head_dim != 64) {   // every Whisper checkpoint has 64-wide heads (n_state / 64 heads)
        set_error("wt_qk_rows_batch: head_dim=%d unsupported (Whisper heads are 64 wide)", head_dim);
        return WT_E_UNSUPPORTED;
    }
    if (n_sel == 0 || n_batch == 0) return WT_OK;
    QkLayers L = {};
    for (int l = 0; l < n_layers; ++l) {
        if (!q_layers[l] || !k_layers[l]) {
            set_error("wt_qk_rows_batch: layer %d has a null projection", l);
            return WT_E_BADARG;
        }
        L.q[l] = q_layers[l];
        L.k[l] = k_layers[l];
    }
    const dim3 grid((n_ctx + 255) / 256, n_sel, n_batch), block(256);
#define WT_QKB(ST, DT)                                                                                                   \
    hipLaunchKernelGGL((qk_rows_batch_kernel<ST, DT, 64>), grid, block, 0, st, L, n_q, q_bstride, k_bstride, n_ctx, d_model, \
                       scale, sel_layer, sel_head, sel_slot, row_begin, row_end, ring_index, (DT *)ring, ring_bstride, ring_rows, \
                       ring_row0)
    if (dtype == WT_DTYPE_F32 && ring_dtype == WT_DTYPE_F32) WT_QKB(float, float);
    else if (dtype == WT_DTYPE_F32 && ring_dtype == WT_DTYPE_F16) WT_QKB(float, __half);
    else if (dtype == WT_DTYPE_F16 && ring_dtype == WT_DTYPE_F32) WT_QKB(__half, float);
    else if (dtype == WT_DTYPE_F16 && ring_dtype == WT_DTYPE_F16) WT_QKB(__half, __half);
    else {
        set_error("wt_qk_rows_batch: dtype %d -> %d", dtype, ring_dtype);
        return WT_E_BADARG;
    }
#undef WT_QKB
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
