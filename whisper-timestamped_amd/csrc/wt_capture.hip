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

}  // namespace wt
