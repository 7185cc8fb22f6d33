// Which alignment units take the fused small-unit tail kernel (wt_small.hip), and what it needs of LDS.  Pure functions of a
// unit's own (T, F) -- host and device agree, and the same unit takes the same path in any batch.
#pragma once
#include "wt_common.h"

namespace wt {

constexpr int WT_SMALL_MAX_T = WT_MAX_TOKENS;   // one lane per token row, ceil(T/64) pipelined waves sweep the DTW
constexpr int WT_SMALL_MAX_LDS = 160 * 1024;    // per workgroup: all of a CU's LDS
constexpr int WT_SMALL_LIGHT_LDS = 32 * 1024;   // units below this share a launch that fits 4-5 workgroups per CU

// floats per row of the skewed LDS matrix (row i holds frame j at column i + j): a multiple of 4 whose quarter is odd
// (16-byte aligned rows, 64 lanes x ds_read_b128 spread over the banks)
__host__ __device__ inline int wt_small_pitch(int T, int F) {
    int p = (F + T + 3) & ~3;
    if ((p & 4) == 0) p += 4;
    return p;
}
// direction planes of the DTW phase (>= 2 blocks of 64 word pairs)
__host__ __device__ inline int wt_small_plane_bytes(int T, int F) {
    const int nw = (T + 63) / 64;
    return ((F + 63 + 31) / 32 + 1) * 64 * nw * 8;
}
// boundary rows between the sweeping waves + parking areas + progress words (wt_dtw_core.h: dtw_bnd_pitch, DUMP)
__host__ __device__ inline int wt_small_bnd_bytes(int T, int F) {
    const int nw = (T + 63) / 64;
    return (nw - 1) * (((F + 64 + 32 + 1) & ~1) + 96) * 8 + 16;
}
constexpr int WT_SMALL_SLACK = 192;             // floats behind the matrix: the last block prefetch of the last rows
__host__ __device__ inline long long wt_small_lds_bytes(int T, int F) {
    return (long long)wt_small_plane_bytes(T, F) + wt_small_bnd_bytes(T, F) +
           ((long long)T * wt_small_pitch(T, F) + WT_SMALL_SLACK) * 4 + 64;
}
__host__ __device__ inline bool wt_small_unit(int T, int F) {
    return T >= 1 && T <= WT_SMALL_MAX_T && F >= 1 && F <= WT_MAX_FRAMES && wt_small_lds_bytes(T, F) <= WT_SMALL_MAX_LDS;
}

}  // namespace wt
