// Fused alignment of SMALL units: local cost + DTW + backtrack in ONE workgroup, the cost matrix never leaves LDS.
//
// The reference's default call shape (transcribe.py:544-557, trust_whisper_timestamps=True) aligns segment by
// segment: one perform_word_alignment (transcribe.py:1428-1793) per Whisper segment, T p50 11 tokens x F p50 144
// frames.  For such units the batched kernels (wt_cost.hip + wt_dtw.hip: rowmean, colnorm, fix00, dtw = up to nine
// launches, the (T,F) matrix written to and read back from HBM twice) are launch- and latency-bound: 0.045 + 0.061 ms
// for 160 real-shape units at 0.04 / 0.004 of the HBM peak (profiles/r2q_bench_kreal.json).  Here one workgroup of four
// waves owns a unit:
//   1. cost rows   wave w takes token rows w, w+4, ...: head_sum_row (wt_cost_core.h: LDS-DMA of each selected head's
//                  row, median-9, softmax, head sum in registers) -> the head mean goes to the unit's LDS matrix
//   2. columns     one thread per frame: sum of squares over tokens in f64 with colnorm_kernel's summation tree (the
//                  matrix is BIT-IDENTICAL to the batched kernels'), normalise, negate, pad mask, unit max
//   3. cost[0,0] = min; optionally the matrix is written to HBM (callers that keep it: disfluency detection, tests)
//   4. DTW         wave 0, one lane per token row, the anti-diagonal sweep of wt_dtw_core.h reading 32-frame blocks of
//                  its row from LDS (the matrix is stored SKEWED, row i shifted right by i, so that at step s every
//                  lane reads column s of its row: 16-byte aligned ds_read_b128), direction planes to LDS
//   5. backtrack   wt_dtw_core.h, planes from LDS -> jumps[T+1] (+ path, distance)
// HBM traffic of a unit = its A*T*F logits in, 4(T+1) bytes out.  A unit qualifies by its own shape alone
// (wt_small_unit: T <= 64 and the LDS it needs), so the same unit takes the same path in any batch.
#include <algorithm>
#include <mutex>

#include "wt_cost_core.h"
#include "wt_dtw_core.h"
#include "wt_small.h"

namespace wt {

typedef float float4v_s __attribute__((ext_vector_type(4)));

// 32 consecutive cost values of this lane's (skewed) row, columns [s0, s0 + 32): eight 16-byte LDS reads
__device__ __forceinline__ void load_blk_lds(const float *rowp, int s0, float (&dst)[BLK]) {
    const float4 *p = reinterpret_cast<const float4 *>(rowp + s0);
#pragma unroll
    for (int k = 0; k < BLK / 4; ++k) {
        const float4 r = p[k];
        dst[4 * k] = r.x; dst[4 * k + 1] = r.y; dst[4 * k + 2] = r.z; dst[4 * k + 3] = r.w;
    }
}

template <int C, typename QT>
__global__ __launch_bounds__(256) void small_align_kernel(const QT *__restrict__ qk, const wt_seg_desc *__restrict__ segs,
                                                          const int32_t *__restrict__ head_idx, int n_heads, float qk_scale,
                                                          float *cost_out, int32_t *__restrict__ jumps,
                                                          int32_t *__restrict__ path_i, int32_t *__restrict__ path_j,
                                                          int32_t *__restrict__ path_len, double *__restrict__ dist, int unit0,
                                                          int f_lo, int f_hi, int need_lo, int need_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BUF = RowBuf<C, QT>::BUF;
    const int unit = unit0 + blockIdx.x;
    const wt_seg_desc d = segs[unit];
    const int T = d.T, F = d.F;
    // block-uniform: the unit belongs to the batched kernels, to another F class or to the other LDS class of this one
    if (!wt_small_unit(T, F) || F <= f_lo || F > f_hi) return;
    const int need = (int)wt_small_lds_bytes(T, F);
    if (need <= need_lo || need > need_hi) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = wt_small_pitch(T, F);
    const int stage_bytes = max(wt_small_stage_bytes(F), wt_small_plane_bytes(T, F));
    // [ row staging (cost phase) | direction planes (DTW phase) ] [ cost matrix, skewed, + 64 floats of slack ] [ reductions ]
    float (*stage)[2][BUF] = reinterpret_cast<float (*)[2][BUF]>(smem);
    uint2 *planes = reinterpret_cast<uint2 *>(smem);
    float *cm = reinterpret_cast<float *>(smem + stage_bytes);
    float *red = cm + (size_t)T * pitch + 64;

    for (int e = tid; e < T * pitch + 64; e += 256) cm[e] = 0.f;   // (cells outside a row's frames must be finite)
    __syncthreads();

    // ---- 1. head mean of every token row -> cm[t][t + f] ----
    for (int t = wave; t < T; t += 4) {
        const QT *row0 = qk + d.qk_offset + (int64_t)t * d.row_stride + d.start_token;
        f2 acc[C / 2];
        head_sum_row<C, QT>(row0, d.head_stride, head_idx, n_heads, F, qk_scale, stage[wave], lane, acc);
        float *dst = cm + (size_t)t * pitch + t + lane * C;
#pragma unroll
        for (int q = 0; q < C / 2; ++q) {
            const int f = lane * C + 2 * q;
            if (f < F) dst[2 * q] = head_mean(acc[q].x, n_heads);
            if (f + 1 < F) dst[2 * q + 1] = head_mean(acc[q].y, n_heads);
        }
    }
    __syncthreads();

    // ---- 2. per frame: L2 norm over tokens (f64, colnorm_kernel's tree: rows t = w mod 16 summed in order, then the
    //         sixteen partial sums in order), normalise, negate, pad mask; the unit's largest w / norm ----
    float umax = 0.f;
    for (int f = tid; f < F; f += 256) {
        const bool masked_col = d.pad_from > 0 && f >= d.pad_from;   // 0 = no mask, like the reference's `if max_duration:`
        float *col = cm + f;
        double part[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) part[w] = 0.0;
        float m = 0.f;
        for (int t0 = 0; t0 < T; t0 += 16) {
#pragma unroll
            for (int w = 0; w < 16; ++w) {
                const int t = t0 + w;
                if (t < T) {
                    const float v = col[(size_t)t * (pitch + 1)];
                    part[w] += (double)v * (double)v;
                    if (!masked_col || t == T - 1) m = fmaxf(m, v);
                }
            }
        }
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += part[w];
        const float norm = sqrtf((float)tot);
        umax = fmaxf(umax, m / norm);   // max_t(w/norm) == max_t(w)/norm: IEEE division is monotone
        for (int t = 0; t < T; ++t) {
            float *c = col + (size_t)t * (pitch + 1);
            *c = (masked_col && t < T - 1) ? 0.f : -(*c / norm);
        }
    }
    umax = wave_max(umax);
    if (lane == 0) red[wave] = umax;
    __syncthreads();
    // ---- 3. cost[0,0] = min(cost) (transcribe.py:1568); the matrix to HBM for callers that keep it ----
    if (tid == 0) cm[0] = -fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    if (cost_out) {
        float *out = cost_out + d.cost_offset;
        for (int t = 0; t < T; ++t)
            for (int f = tid; f < F; f += 256) out[(size_t)t * F + f] = cm[(size_t)t * pitch + t + f];
    }
    if (wave != 0) return;   // (no barrier below: the staging area is free for the planes since the barrier above)

    // ---- 4. DTW sweep on wave 0: lane = token row ----
    const int i = lane;
    const int nsteps = F + T - 1;
    const float *rowp = cm + (size_t)min(i, T - 1) * pitch;   // lanes >= T repeat the last row: finite, they feed no valid cell
    const double INF = __builtin_inf();
    double g = INF, u0 = INF, u1 = (i == 0) ? 0.0 : INF;   // first diagonal: 0 + lm[0,0] reproduces cm[0,0] = lm[0,0]
    double gfinal = 0.0;
    const int sfinal = F - 1 + lane;
    uint32_t wa = 0, wb = 0;
    float bufA[BLK], bufB[BLK];
    double edge[BLK];        // (unused: a single wave has no producer above it)
    load_blk_lds(rowp, 0, bufA);
    uint2 *pword = planes + lane;
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    auto sweep = [&](auto dist_c) __attribute__((always_inline)) {
        constexpr bool DIST = decltype(dist_c)::value;
        auto block = [&](const float (&cur)[BLK], float (&nxt)[BLK], int s0, auto first_c) __attribute__((always_inline)) {
            load_blk_lds(rowp, s0 + BLK, nxt);     // the next block's costs (the slack behind the matrix covers the last one)
            asm volatile("" ::: "memory");
            sweep_block<false, false, DIST, decltype(first_c)::value, false>(cur, g, u0, u1, edge, wa, wb, nullptr, s0, sfinal, gfinal);
            *pword = make_uint2(wa, wb);
            pword += 64;
        };
        block(bufA, bufB, 0, yes{});
        for (int s0 = BLK; s0 < nsteps; s0 += 2 * BLK) {
            block(bufB, bufA, s0, no{});
            if (s0 + BLK < nsteps) block(bufA, bufB, s0 + BLK, no{});
        }
    };
    if (dist) sweep(yes{}); else sweep(no{});
    if (dist && i == T - 1) dist[unit] = gfinal;

    // ---- 5. backtrack + jumps ----
    backtrack_unit([&](int k) { return planes[k]; }, T, F, 64, lane, unit, jumps + d.jumps_offset,
                   (path_i && path_j) ? path_i + d.path_offset : nullptr, (path_i && path_j) ? path_j + d.path_offset : nullptr,
                   path_len);
}

template <int C, typename QT>
static int launch_small(const QT *qk, const wt_seg_desc *segs_dev, int unit0, int n, int f_lo, int f_hi, int need_lo,
                        int need_hi, const int32_t *head_idx, int n_heads, float qk_scale, float *cost, int32_t *jumps,
                        int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    static std::mutex mu;
    static bool attr_set[64] = {false};   // function attributes are per (instantiation, device)
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        set_error("wt_align_batch: device ordinal %d out of range", dev);
        return WT_E_UNSUPPORTED;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[dev]) {
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_align_kernel<C, QT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL((small_align_kernel<C, QT>), dim3(n), dim3(256), (size_t)need_hi, st, qk, segs_dev, head_idx, n_heads,
                       qk_scale, cost, jumps, path_i, path_j, path_len, dist, unit0, f_lo, f_hi, need_lo, need_hi);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// Launch plan: units are grouped by the row instantiation their F needs (C = 4, 8, 16 or 28 elements per lane) and,
// inside a group, by LDS appetite ("light" <= WT_SMALL_LIGHT_LDS: several workgroups per CU; "heavy": the rest), each
// (group, appetite) present = one launch over the group's unit range with the largest LDS of its members.
template <typename QT>
static int align_small_t(const QT *qk, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                         const int32_t *head_idx, int n_heads, float qk_scale, float *cost, int32_t *jumps, int32_t *path_i,
                         int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    static const int f_edges[5] = {0, 256, 512, 1024, WT_MAX_FRAMES};
    for (int c = 0; c < 4; ++c) {
        int lo = -1, hi = -1, light_max = 0, heavy_max = 0;
        for (int s = 0; s < n_seg; ++s) {
            const wt_seg_desc &d = segs_host[s];
            if (!wt_small_unit(d.T, d.F) || d.F <= f_edges[c] || d.F > f_edges[c + 1]) continue;
            if (lo < 0) lo = s;
            hi = s;
            const int need = (int)wt_small_lds_bytes(d.T, d.F);
            if (need <= WT_SMALL_LIGHT_LDS) light_max = std::max(light_max, need);
            else heavy_max = std::max(heavy_max, need);
        }
        if (lo < 0) continue;
        for (int pass = 0; pass < 2; ++pass) {
            const int need_lo = pass == 0 ? 0 : WT_SMALL_LIGHT_LDS, need_hi = pass == 0 ? light_max : heavy_max;
            if (need_hi == 0) continue;
            int rc;
#define WT_SMALL_CASE(CI, CC)                                                                                              \
    case CI:                                                                                                               \
        rc = launch_small<CC, QT>(qk, segs_dev, lo, hi - lo + 1, f_edges[c], f_edges[c + 1], need_lo, need_hi, head_idx,    \
                                  n_heads, qk_scale, cost, jumps, path_i, path_j, path_len, dist, st);                     \
        break;
            switch (c) {
                WT_SMALL_CASE(0, 4)
                WT_SMALL_CASE(1, 8)
                WT_SMALL_CASE(2, 16)
                default:
                WT_SMALL_CASE(3, 28)
            }
#undef WT_SMALL_CASE
            if (rc) return rc;
        }
    }
    return WT_OK;
}

int align_small(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                const int32_t *head_idx, int n_heads, float qk_scale, float *cost, int32_t *jumps, int32_t *path_i,
                int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    if (qk_dtype == WT_DTYPE_F32)
        return align_small_t((const float *)qk, segs_host, segs_dev, n_seg, head_idx, n_heads, qk_scale, cost, jumps, path_i,
                             path_j, path_len, dist, st);
    if (qk_dtype == WT_DTYPE_F16)
        return align_small_t((const __half *)qk, segs_host, segs_dev, n_seg, head_idx, n_heads, qk_scale, cost, jumps, path_i,
                             path_j, path_len, dist, st);
    set_error("wt_align_batch: qk_dtype=%d", qk_dtype);
    return WT_E_BADARG;
}

}  // namespace wt
