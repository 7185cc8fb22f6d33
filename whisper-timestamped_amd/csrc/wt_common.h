// Shared device/host helpers for libwtalign (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wtalign.h"

#define WT_WAVE 64

namespace wt {

// ---- error plumbing (host) -------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
#define WT_HIP(call)                                        \
    do {                                                    \
        hipError_t _e = (call);                             \
        if (_e != hipSuccess) return ::wt::hip_fail(_e, #call); \
    } while (0)

// scratch arena of (current device, stream): `bytes` of device memory, grown on demand
int scratch(hipStream_t st, size_t bytes, void **out);

// ---- wave-level primitives (device) ---------------------------------------
// Cross-lane LDS hand-off inside ONE wave: DS ops of a wave execute in issue
// order, so only the compiler must be kept from reordering them.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// DPP butterflies inside each row of 16 lanes, then 4 readlanes: no LDS-crossbar (ds_bpermute) traffic.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rows_combine_max(float v) {
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_max_dpp(float v) {  // result is wave-uniform
    v = fmaxf(v, dpp_mov<0xB1>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));  // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));  // row_mirror
    return rows_combine_max(v);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {  // fixed summation tree, wave-uniform result
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}

// s_waitcnt vmcnt(0) only (gfx9 encoding: vmcnt[3:0]|[15:14], expcnt[6:4], lgkmcnt[11:8])
__device__ __forceinline__ void wait_vmcnt0() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// lane i receives lane i-1's value; lane 0 keeps `lane0` (DPP wave_shr:1, GFX9)
__device__ __forceinline__ double wave_shr1(double v, double lane0) {
    union { double d; int i[2]; } s, o, r;
    s.d = v;
    o.d = lane0;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xf, 0xf, false);
    return r.d;
}

// scipy.ndimage 'reflect' (= numpy 'symmetric', half-sample symmetric, period 2n)
__device__ __forceinline__ int reflect_index(int i, int n) {
    int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

}  // namespace wt
