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
//   4. DTW         ceil(T/64) waves pipelined through LDS boundary rows, one lane per token row, the anti-diagonal sweep of wt_dtw_core.h reading 32-frame blocks of
//                  its row from LDS (the matrix is stored SKEWED, row i shifted right by i, so that at step s every
//                  lane reads column s of its row: 16-byte aligned ds_read_b128), direction planes to LDS
//   5. backtrack   wt_dtw_core.h, planes from LDS -> jumps[T+1] (+ path, distance)
// The matrix crosses HBM once in each direction instead of five times, and a batch of small units is TWO launches
// (rowmean + this) instead of up to nine.  A unit qualifies by its own shape alone (wt_small_unit: T <= 256 and the 160 KB of LDS
// its matrix, planes and boundary rows need), so the same unit takes the same path in any batch.
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

// DIST: the caller wants alignment.distance.  MULTI: units of more than 64 token rows (several pipelined waves) are
// served; the <*, false> instantiations carry the single-wave sweep only (fewer registers: two or three workgroups per CU).
template <bool DIST, bool MULTI>
__global__ __launch_bounds__(256) void small_tail_kernel(const wt_seg_desc *__restrict__ segs, float *cost, int write_cost,
                                                         int32_t *__restrict__ jumps, int32_t *__restrict__ path_i,
                                                         int32_t *__restrict__ path_j, int32_t *__restrict__ path_len,
                                                         double *__restrict__ dist, int unit0, int need_lo, int need_hi,
                                                         int nw_lo, int nw_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int unit = unit0 + blockIdx.x;
    const wt_seg_desc d = segs[unit];
    const int T = d.T, F = d.F;
    if (!wt_small_unit(T, F)) return;   // block-uniform: the unit belongs to the batched kernels
    const int need = (int)wt_small_lds_bytes(T, F);
    if (need <= need_lo || need > need_hi) return;   // ... or to another launch of this kernel
    if ((T + 63) / 64 < nw_lo || (T + 63) / 64 > nw_hi) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: lives in an SGPR
    const int pitch = wt_small_pitch(T, F);
    const int nw = (T + 63) >> 6;                // sweeping waves: wave w owns token rows 64w .. 64w+63
    const int rowsP = nw * 64;                   // rows of one block of plane words
    const int bpitch = dtw_bnd_pitch(F);
    // [ direction planes ] [ boundary rows, parking areas, progress words ] [ cost matrix, skewed, + slack ] [ reductions ]
    uint2 *planes = reinterpret_cast<uint2 *>(smem);
    double *bnd = reinterpret_cast<double *>(smem + wt_small_plane_bytes(T, F));   // [nw-1][bpitch], bnd[w][64 + j]
    double *park = bnd + (size_t)(nw - 1) * bpitch;                                // [nw-1][DUMP]
    int *prog = reinterpret_cast<int *>(park + (size_t)(nw - 1) * DUMP);           // [nw-1]
    float *cm = reinterpret_cast<float *>(smem + wt_small_plane_bytes(T, F) + wt_small_bnd_bytes(T, F));
    float *red = cm + (size_t)T * pitch + WT_SMALL_SLACK;
    float *unit_cost = cost + d.cost_offset;
    if (tid < nw) prog[tid] = 0;

    // ---- 1. the head-mean matrix -> cm[t][t + f].  Cells outside a row's frames must be finite: everything is zeroed
    //         first; then the T*F values are fetched in batches of eight independent, coalesced loads per thread ----
    for (int e = tid; e < T * pitch + WT_SMALL_SLACK; e += 256) cm[e] = 0.f;
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
    if (lane == 0) red[tid >> 6] = umax;
    __syncthreads();
    // ---- 3. cost[0,0] = min(cost) (transcribe.py:1568); the matrix to HBM for callers that keep it ----
    if (tid == 0) cm[0] = -fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    if (write_cost)
        for (int t = tid >> 6; t < T; t += 4)
            for (int f = lane; f < F; f += 64) unit_cost[(size_t)t * F + f] = cm[(size_t)t * pitch + t + f];

    // ---- 4. DTW: lane = token row, wave w sweeps rows 64w..64w+63 pipelined behind wave w-1 through an LDS boundary row
    //         (the scheme of wt_dtw.hip's dtw_kernel; here the cost blocks come from LDS) ----
    if (wave < nw) {
        const int i = wave * 64 + lane;
        const int nsteps = nw == 1 ? F + T - 1 : F + 63;   // (one wave: nothing happens behind the last row's last frame)
        // skewed row: frame j of row i sits at column i + j = 64w + s (s = this wave's step j + lane); lanes beyond T
        // repeat the last row: finite values, they feed no valid cell
        const float *rowp = cm + (size_t)min(i, T - 1) * pitch + 64 * wave;
        const double INF = __builtin_inf();
        double g = INF, u0 = INF, u1 = (i == 0) ? 0.0 : INF;   // first diagonal: 0 + lm[0,0] reproduces cm[0,0] = lm[0,0]
        double gfinal = 0.0;
        const int sfinal = F - 1 + lane;
        uint32_t wa = 0, wb = 0;
        float bufA[BLK], bufB[BLK];
        load_blk_lds(rowp, 0, bufA);
        uint2 *pword = planes + i;
        const bool producer = wave < nw - 1;
        const int pw = producer ? wave : 0;
        double *pub = (lane == 63) ? bnd + (size_t)pw * bpitch + 1 : park + (size_t)pw * DUMP + lane;
        const int pubinc = (lane == 63) ? BLK : 0;
        const double *erow = bnd + (size_t)(wave > 0 ? wave - 1 : 0) * bpitch;
        using yes = std::integral_constant<bool, true>;
        using no = std::integral_constant<bool, false>;
        auto sweep = [&](auto edge_c, auto publish_c) __attribute__((always_inline)) {
            constexpr bool EDGE = decltype(edge_c)::value, PUBLISH = decltype(publish_c)::value;
            auto block = [&](const float (&cur)[BLK], float (&nxt)[BLK], int s0, auto first_c) __attribute__((always_inline)) {
                load_blk_lds(rowp, s0 + BLK, nxt);     // the next block's costs (the slack behind the matrix covers the last one)
                asm volatile("" ::: "memory");
                double edge[BLK];
                if (EDGE) {
                    const int need = min(s0 + BLK, F);
                    while (__hip_atomic_load(&prog[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                        __builtin_amdgcn_s_sleep(1);
                    const double2 *e2 = reinterpret_cast<const double2 *>(erow + min(64 + s0, bpitch - BLK));
#pragma unroll
                    for (int k = 0; k < BLK / 2; ++k) {
                        const double2 v = e2[k];
                        edge[2 * k] = v.x;
                        edge[2 * k + 1] = v.y;
                    }
                }
                sweep_block<EDGE, PUBLISH, DIST, decltype(first_c)::value, false>(cur, g, u0, u1, edge, wa, wb, pub, s0, sfinal, gfinal);
                *pword = make_uint2(wa, wb);
                pword += rowsP;
                if (PUBLISH) {
                    pub += pubinc;
                    if (lane == 0) {
                        const int done = min(max(s0 + BLK - 63, 0), F);  // frames of row 64w+63 finished so far
                        __hip_atomic_store(&prog[wave], done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            };
            block(bufA, bufB, 0, std::integral_constant<bool, !EDGE>{});
            for (int s0 = BLK; s0 < nsteps; s0 += 2 * BLK) {
                block(bufB, bufA, s0, no{});
                if (s0 + BLK < nsteps) block(bufA, bufB, s0 + BLK, no{});
            }
        };
        // the role of a wave (a producer above / a consumer below) is decided once, outside the sweep
        if (!MULTI || nw == 1) sweep(no{}, no{});
        else if (wave == 0) sweep(no{}, yes{});
        else if (producer) sweep(yes{}, yes{});
        else sweep(yes{}, no{});
        if (DIST && i == T - 1) dist[unit] = gfinal;
    }
    __syncthreads();        // every wave's plane words are in LDS
    if (wave != 0) return;

    // ---- 5. backtrack + jumps ----
    backtrack_unit([&](int k) { return planes[k]; }, T, F, rowsP, lane, unit, jumps + d.jumps_offset,
                   (path_i && path_j) ? path_i + d.path_offset : nullptr, (path_i && path_j) ? path_j + d.path_offset : nullptr,
                   path_len);
}

template <bool DIST, bool MULTI>
static int launch_tail_t(const wt_seg_desc *segs_dev, int unit0, int n, int need_lo, int need_hi, int nw_lo, int nw_hi, float *cost,
                         int write_cost, int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist,
                         hipStream_t st) {
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
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(small_tail_kernel<DIST, MULTI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    hipLaunchKernelGGL((small_tail_kernel<DIST, MULTI>), dim3(n), dim3(256), (size_t)need_hi, st, segs_dev, cost, write_cost, jumps,
                       path_i, path_j, path_len, dist, unit0, need_lo, need_hi, nw_lo, nw_hi);
    WT_HIP(hipGetLastError());
    return WT_OK;
}
static int launch_tail(bool multi, const wt_seg_desc *segs_dev, int unit0, int n, int need_lo, int need_hi, int nw_lo, int nw_hi,
                       float *cost, int write_cost, int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len,
                       double *dist, hipStream_t st) {
#define WT_TAIL(D, M) launch_tail_t<D, M>(segs_dev, unit0, n, need_lo, need_hi, nw_lo, nw_hi, cost, write_cost, jumps, path_i, \
                                          path_j, path_len, dist, st)
    if (dist) return multi ? WT_TAIL(true, true) : WT_TAIL(true, false);
    return multi ? WT_TAIL(false, true) : WT_TAIL(false, false);
#undef WT_TAIL
}

// A launch lasts as long as its longest DTW chain, so splitting a batch adds chains up: a batch with no more small units
// than the chip has CUs is ONE launch over the range of its small units (largest LDS appetite and, when a unit of more
// than 64 rows is among them, the multi-wave instantiation for all).  With <= 256 workgroups every unit has a CU of its
// own whatever LDS the launch reserves, so the shared sizing costs nothing in occupancy; what the light units pay in a
// mixed launch is the multi-wave body's per-step overhead (progress-word test), which is why RESULTS never depend on the
// batch but a light unit's share of the launch time can (accepted, documented).  A larger batch is split by what limits how many
// workgroups share a CU: single-wave units apart from multi-wave ones (registers), "light" units (<= WT_SMALL_LIGHT_LDS)
// apart from heavy ones (LDS).
int align_small_tail(const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, float *cost, bool keep_cost,
                     int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    int lo = -1, hi = -1, n_small = 0;
    int need_max[2][2] = {{0, 0}, {0, 0}};   // [multi-wave][heavy]
    for (int s = 0; s < n_seg; ++s) {
        const wt_seg_desc &d = segs_host[s];
        if (!wt_small_unit(d.T, d.F)) continue;
        if (lo < 0) lo = s;
        hi = s;
        ++n_small;
        const int need = (int)wt_small_lds_bytes(d.T, d.F);
        int &m = need_max[d.T > 64][need > WT_SMALL_LIGHT_LDS];
        m = std::max(m, need);
    }
    if (lo < 0) return WT_OK;
    const int wc = keep_cost ? 1 : 0, n = hi - lo + 1;
    const bool any_multi = need_max[1][0] || need_max[1][1];
    if (n_small <= 256) {
        const int need = std::max(std::max(need_max[0][0], need_max[0][1]), std::max(need_max[1][0], need_max[1][1]));
        return launch_tail(any_multi, segs_dev, lo, n, 0, need, 1, 4, cost, wc, jumps, path_i, path_j, path_len, dist, st);
    }
    for (int multi = 0; multi < 2; ++multi)
        for (int heavy = 0; heavy < 2; ++heavy) {
            if (!need_max[multi][heavy]) continue;
            const int rc = launch_tail(multi != 0, segs_dev, lo, n, heavy ? WT_SMALL_LIGHT_LDS : 0, need_max[multi][heavy],
                                       multi ? 2 : 1, multi ? 4 : 1, cost, wc, jumps, path_i, path_j, path_len, dist, st);
            if (rc) return rc;
        }
    return WT_OK;
}

}  // namespace wt
