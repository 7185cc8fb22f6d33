// Fused tail of the alignment of SMALL units: column norm + cost[0,0] + DTW + backtrack in ONE workgroup, the cost matrix
// in LDS.
//
// The reference's default call shape (transcribe.py:544-557, trust_whisper_timestamps=True) aligns segment by
// segment: one perform_word_alignment (transcribe.py:1428-1793) per Whisper segment, T p50 11 tokens x F p50 144
// frames.  For such units the batched kernels (rowmean, colnorm, fix00, dtw: up to nine launches, the (T,F) matrix
// written, read-modified-written twice and read again) are launch- and latency-bound: 0.045 + 0.061 ms for 160
// real-shape units at 0.04 / 0.004 of the HBM peak (profiles/r2q_bench_kreal.json).  Here the rows still come from the
// batched rowmean_kernel (thousands of waves in flight hide the HBM latency of the A*T row fetches; one workgroup
// fetching its unit's rows itself was measured 4x slower: a wave has one or two rows in flight) and ONE workgroup of
// four waves does everything after it:
//   1. load        the unit's head-mean matrix (T*F fp32, what rowmean wrote) HBM -> LDS, every load in flight at once
//   2. columns     one thread per frame: sum of squares over tokens in f64 with colnorm_kernel's summation tree (the
//                  matrix is BIT-IDENTICAL to the batched kernels'), normalise, negate, pad mask, unit max
//   3. cost[0,0] = min; optionally the matrix goes back to HBM (callers that keep it: disfluency detection, tests)
//   4. DTW         wave 0, one lane per token row, the anti-diagonal sweep of wt_dtw_core.h reading 32-frame blocks of
//                  its row from LDS (the matrix is stored SKEWED, row i shifted right by i, so that at step s every
//                  lane reads column s of its row: 16-byte aligned ds_read_b128), direction planes to LDS
//   5. backtrack   wt_dtw_core.h, planes from LDS -> jumps[T+1] (+ path, distance)
// The matrix crosses HBM once in each direction instead of five times, and a batch of small units is TWO launches
// (rowmean + this) instead of up to nine.  A unit qualifies by its own shape alone (wt_small_unit: T <= 64 and the LDS
// it needs), so the same unit takes the same path in any batch.
#include <algorithm>
#include <mutex>

#include "wt_dtw_core.h"
#include "wt_small.h"

namespace wt {

// 32 consecutive cost values of this lane's (skewed) row, columns [s0, s0 + 32): eight 16-byte LDS reads
__device__ __forceinline__ void load_blk_lds(const float *rowp, int s0, float (&dst)[BLK]) {
    const float4 *p = reinterpret_cast<const float4 *>(rowp + s0);
#pragma unroll
    for (int k = 0; k < BLK / 4; ++k) {
        const float4 r = p[k];
        dst[4 * k] = r.x; dst[4 * k + 1] = r.y; dst[4 * k + 2] = r.z; dst[4 * k + 3] = r.w;
    }
}

__global__ __launch_bounds__(256) void small_tail_kernel(const wt_seg_desc *__restrict__ segs, float *cost, int write_cost,
                                                         int32_t *__restrict__ jumps, int32_t *__restrict__ path_i,
                                                         int32_t *__restrict__ path_j, int32_t *__restrict__ path_len,
                                                         double *__restrict__ dist, int unit0, int need_lo, int need_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int unit = unit0 + blockIdx.x;
    const wt_seg_desc d = segs[unit];
    const int T = d.T, F = d.F;
    if (!wt_small_unit(T, F)) return;   // block-uniform: the unit belongs to the batched kernels
    const int need = (int)wt_small_lds_bytes(T, F);
    if (need <= need_lo || need > need_hi) return;   // ... or to the other launch of this kernel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pitch = wt_small_pitch(T, F);
    // [ direction planes ] [ cost matrix, skewed, + 64 floats of slack ] [ reductions ]
    uint2 *planes = reinterpret_cast<uint2 *>(smem);
    float *cm = reinterpret_cast<float *>(smem + wt_small_plane_bytes(T, F));
    float *red = cm + (size_t)T * pitch + 64;
    float *unit_cost = cost + d.cost_offset;

    // ---- 1. the head-mean matrix -> cm[t][t + f].  Cells outside a row's frames must be finite: everything is zeroed
    //         first; then the T*F values are fetched in batches of eight independent, coalesced loads per thread ----
    for (int e = tid; e < T * pitch + 64; e += 256) cm[e] = 0.f;
    __syncthreads();
    const int n_el = T * F;
    for (int e0 = tid; e0 < n_el; e0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + 256 * k;
            v[k] = e < n_el ? unit_cost[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + 256 * k;
            if (e < n_el) {
                const int t = (int)((unsigned)e / (unsigned)F);
                cm[(size_t)t * (pitch + 1) + (e - t * F)] = v[k];
            }
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
    if (write_cost)
        for (int t = wave; t < T; t += 4)
            for (int f = lane; f < F; f += 64) unit_cost[(size_t)t * F + f] = cm[(size_t)t * pitch + t + f];
    if (wave != 0) return;   // (no barrier below)

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

static int launch_tail(const wt_seg_desc *segs_dev, int unit0, int n, int need_lo, int need_hi, float *cost, int write_cost,
                       int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    static std::mutex mu;
    static bool attr_set[64] = {false};   // function attributes are per device
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        set_error("wt_align_batch: device ordinal %d out of range", dev);
        return WT_E_UNSUPPORTED;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[dev]) {
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_tail_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL(small_tail_kernel, dim3(n), dim3(256), (size_t)need_hi, st, segs_dev, cost, write_cost, jumps, path_i, path_j,
                       path_len, dist, unit0, need_lo, need_hi);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// ONE launch over the range of the small units, with the largest LDS appetite among them -- a launch lasts as long as
// its longest DTW chain, so splitting a batch by size would add the chains up.  Only a batch with more small units than
// the chip has CUs is split in two ("light" units <= WT_SMALL_LIGHT_LDS share a CU four or five at a time).
int align_small_tail(const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, float *cost, bool keep_cost,
                     int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    int lo = -1, hi = -1, n_small = 0, light_max = 0, heavy_max = 0;
    for (int s = 0; s < n_seg; ++s) {
        const wt_seg_desc &d = segs_host[s];
        if (!wt_small_unit(d.T, d.F)) continue;
        if (lo < 0) lo = s;
        hi = s;
        ++n_small;
        const int need = (int)wt_small_lds_bytes(d.T, d.F);
        if (need <= WT_SMALL_LIGHT_LDS) light_max = std::max(light_max, need);
        else heavy_max = std::max(heavy_max, need);
    }
    if (lo < 0) return WT_OK;
    const int wc = keep_cost ? 1 : 0;
    if (n_small <= 256 || light_max == 0 || heavy_max == 0)
        return launch_tail(segs_dev, lo, hi - lo + 1, 0, std::max(light_max, heavy_max), cost, wc, jumps, path_i, path_j, path_len,
                           dist, st);
    int rc = launch_tail(segs_dev, lo, hi - lo + 1, 0, light_max, cost, wc, jumps, path_i, path_j, path_len, dist, st);
    if (rc) return rc;
    return launch_tail(segs_dev, lo, hi - lo + 1, WT_SMALL_LIGHT_LDS, heavy_max, cost, wc, jumps, path_i, path_j, path_len, dist, st);
}

}  // namespace wt
