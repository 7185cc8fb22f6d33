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

}  // namespace wt
