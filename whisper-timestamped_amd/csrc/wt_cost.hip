// Local-cost construction for the DTW word alignment, batched over units.
//
// Replaces the numerics of /root/reference/whisper_timestamped/transcribe.py
//   :1540-1545  slice frames + select alignment heads
//   :1546       scipy.ndimage.median_filter(w, (1,1,9))   ('reflect' edges)
//   :1547       softmax over the F-frame window (* qk_scale first)
//   :1548       mean over heads
//   :1549       divide by the per-frame L2 norm over tokens
//   :1550       negate (the widening to double happens in the DTW kernel)
//   :1561-1565  padding mask (absolute-index quirk carried by wt_seg_desc.pad_from)
//   :1568       cost[0,0] = cost.min()
//
// Kernel 1 (rowmean): ONE WAVE per (unit, token row).  For each selected head
//   the row (F <= 1792 fp32 logits) is copied ONCE from HBM straight into LDS
//   (global_load_lds, double-buffered: head a+1 streams in while head a is
//   being consumed), re-read as one contiguous chunk per lane (+4 halo each
//   side, ds_read_b128), and everything else happens in registers:
//   median-of-9 with the 3x3 sorted-column identity on v_min3/v_med3/v_max3
//   (7 VALU ops per element), wave-wide max / sum by DPP butterflies, exp on
//   v_exp_f32 of ONE rounded product, softmax tail on packed fp32 instructions,
//   accumulate over heads.
//   No workgroup barrier anywhere.  Algorithmic HBM bytes: A*T*F*4 read +
//   T*F*4 written.
// Kernel 2 (colnorm): per 64-column tile, column sum of squares over tokens
//   (f64 accumulate), in-place normalise/negate/mask, per-unit min via one
//   atomicMax on the magnitude bits (the word is zeroed by kernel 1: no memset
//   node).  Kernel 3 (fix00): cost[0,0] = min.
#include "wt_cost_core.h"
#include "wt_small.h"

namespace wt {

// One token row of one unit: the head mean of softmax(median9(.)) -> cost[t, :].  ONE WAVE; `lds` = this wave's NB row
// buffers (wt_cost_core.h).  C = elements per lane; an instantiation can serve any F <= C*64.
template <int C, typename QT, int NB = 2, int MW = 9>
__device__ __forceinline__ void rowmean_row(const QT *__restrict__ qk, const wt_seg_desc &d, int t, const int32_t *__restrict__ head_idx,
                                            int n_heads, float qk_scale, float *__restrict__ cost, float (*lds)[RowBuf<C, QT>::BUF],
                                            int lane) {
    constexpr int CAP = C * 64;
    constexpr int BUF = RowBuf<C, QT>::BUF;
    const int F = d.F;
    const QT *row0 = qk + d.qk_offset + (int64_t)t * d.row_stride + d.start_token;
    f2 acc[C / 2];
    head_sum_row<C, QT, NB, MW>(row0, d.head_stride, head_idx, n_heads, F, qk_scale, lds, lane, acc);

    // mean over heads (torch CPU: sum then div), back through LDS for a coalesced store
    const float nh = (float)n_heads;
    float *xs = &lds[0][0];   // (fp16: the transposition spans both row buffers -- 2 * BUF >= CAP floats; all copies have landed)
    static_assert(2 * BUF >= CAP, "the output transposition fits the wave's two row buffers");
    wave_lds_fence();
    float4 *op = reinterpret_cast<float4 *>(xs + lane * C);
    if ((n_heads & (n_heads - 1)) == 0) {  // power of two: x * (1/n) == x / n exactly
        const float rn = 1.0f / nh;
#pragma unroll
        for (int k = 0; k < C / 4; ++k)
            op[k] = make_float4(acc[2 * k].x * rn, acc[2 * k].y * rn, acc[2 * k + 1].x * rn, acc[2 * k + 1].y * rn);
    } else {
#pragma unroll
        for (int k = 0; k < C / 4; ++k)
            op[k] = make_float4(acc[2 * k].x / nh, acc[2 * k].y / nh, acc[2 * k + 1].x / nh, acc[2 * k + 1].y / nh);
    }
    wave_lds_fence();
    float *out = cost + d.cost_offset + (int64_t)t * F;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const int f = k * 64 + lane;
        if (f < F) out[f] = xs[f];
    }
}

// The launch of ONE F class (or of classes 0 + 1 together: plan_groups).
// EXACT: every unit of the launch has (C-4)*64 < F <= C*64 (single-class launch): the compiler is told, and drops the
// per-chunk guards of the first C-4 row chunks (measured: 0.106 vs 0.124 ms on the K-full cost stage).
template <int C, typename QT, bool EXACT>
__global__ __launch_bounds__(256) void rowmean_kernel(const QT *__restrict__ qk, const wt_seg_desc *__restrict__ segs,
                                                      const int32_t *__restrict__ head_idx, int n_heads, float qk_scale,
                                                      float *__restrict__ cost, unsigned *__restrict__ segstate, int unit0,
                                                      int f_lo, int f_hi) {
    __shared__ __attribute__((aligned(16))) float lds[4][2][RowBuf<C, QT>::BUF];   // per wave: double-buffered row (wt_cost_core.h)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int unit = unit0 + blockIdx.y;
    const wt_seg_desc d = segs[unit];
    const int F = d.F;
    const int t = blockIdx.x * 4 + wave;
    if (F <= f_lo || F > f_hi || t >= d.T) return;  // wave-uniform; the host guarantees f_hi <= CAP
    if (EXACT) __builtin_assume(F > (C - 4) * 64);
    if (t == 0 && lane == 0) segstate[unit] = 0u;  // per-unit max |cost| bits for colnorm (saves a memset node)
    rowmean_row<C, QT>(qk, d, t, head_idx, n_heads, qk_scale, cost, lds[wave], lane);
}

// The same launch for the other median widths of the seam (transcribe.py:1439 medfilt_width; odd, < 9): the width is a
// launch argument, every width its own body.  Not a tuned path (no caller of the reference passes anything but 9): one
// pipelined launch per F class, whatever the batch size.
template <int C, typename QT>
__global__ __launch_bounds__(256) void rowmean_medw_kernel(const QT *__restrict__ qk, const wt_seg_desc *__restrict__ segs,
                                                           const int32_t *__restrict__ head_idx, int n_heads, float qk_scale,
                                                           float *__restrict__ cost, unsigned *__restrict__ segstate, int unit0,
                                                           int f_lo, int f_hi, int medfilt_width) {
    __shared__ __attribute__((aligned(16))) float lds[4][2][RowBuf<C, QT>::BUF];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int unit = unit0 + blockIdx.y;
    const wt_seg_desc d = segs[unit];
    const int F = d.F;
    const int t = blockIdx.x * 4 + wave;
    if (F <= f_lo || F > f_hi || t >= d.T) return;
    if (t == 0 && lane == 0) segstate[unit] = 0u;
    switch (medfilt_width) {
        case 1: rowmean_row<C, QT, 2, 1>(qk, d, t, head_idx, n_heads, qk_scale, cost, lds[wave], lane); break;
        case 3: rowmean_row<C, QT, 2, 3>(qk, d, t, head_idx, n_heads, qk_scale, cost, lds[wave], lane); break;
        case 5: rowmean_row<C, QT, 2, 5>(qk, d, t, head_idx, n_heads, qk_scale, cost, lds[wave], lane); break;
        default: rowmean_row<C, QT, 2, 7>(qk, d, t, head_idx, n_heads, qk_scale, cost, lds[wave], lane); break;
    }
}

// A SMALL batch (the reference's per-segment units: a handful of segments per 30 s window, any mix of lengths) is bound by
// launches and by latency: every rowmean launch lasts its eight sequential head fetches (~12 us) however few rows it
// has.  This kernel serves EVERY class in one launch -- each workgroup runs the body of its own unit's class: the same
// code, hence the same bits, as that class's own launch -- and the short classes (F <= 512) fetch the rows of eight heads
// at once (one memory latency per eight heads instead of eight).  It carries the registers and the LDS of its largest
// body, which is why it stops at F <= 1024 (four workgroups per CU) and why large batches keep one pipelined launch per class.
constexpr int ANY_MAX_CLASS = 3;   // F <= 1024: four workgroups of this kernel share a CU; longer units (rare among per-segment units) keep their class's own launch
constexpr int ANY_WAVE_FLOATS = 8 * RowBuf<4, float>::BUF;   // = 4 buffers of the 8-element class, 2 of the 16-element class
// Which workgroup serves which rows: first[u] = index of unit u's first workgroup (four token rows each), first[n] =
// the grid size.  Passed BY VALUE (kernel arguments): a grid of exactly sum(ceil(T/4)) workgroups instead of
// n_units x ceil(maxT/4) mostly empty ones, each of which would fetch its descriptor only to find it has no rows
// (160 real-shape units: 640 workgroups instead of 2560 -- five dispatch rounds of this kernel's LDS footprint).
constexpr int ANY_MAX_UNITS = 512;
struct WgTable {
    unsigned short first[ANY_MAX_UNITS + 1];
};
template <typename QT>
__global__ __launch_bounds__(256) void rowmean_any_kernel(const QT *__restrict__ qk, const wt_seg_desc *__restrict__ segs,
                                                          const int32_t *__restrict__ head_idx, int n_heads, float qk_scale,
                                                          float *__restrict__ cost, unsigned *__restrict__ segstate, int merge01,
                                                          int n_units, const WgTable table) {
    __shared__ __attribute__((aligned(16))) float raw[4 * ANY_WAVE_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int lo = 0, hi = n_units;                       // the unit whose workgroup range holds blockIdx.x (uniform: scalar unit)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (table.first[mid] <= blockIdx.x) lo = mid; else hi = mid;
    }
    const int unit = lo;
    const wt_seg_desc d = segs[unit];
    const int t = ((int)blockIdx.x - (int)table.first[unit]) * 4 + wave;
    if (t >= d.T || d.F > (ANY_MAX_CLASS + 1) * 256) return;   // wave-uniform (longer units: their class's own launch)
    if (t == 0 && lane == 0) segstate[unit] = 0u;
    int cls = (d.F + 255) / 256 - 1;
    if (cls == 0 && merge01) cls = 1;   // (what a batch with both short classes gets from its shared launch: same bits)
#define WT_ROW_CASE(CI, NBUF)                                                                                        \
    case CI: {                                                                                                        \
        constexpr int BUF = RowBuf<4 * (CI + 1), QT>::BUF;                                                            \
        static_assert(NBUF * BUF <= ANY_WAVE_FLOATS, "the wave's share of the LDS holds its row buffers");            \
        rowmean_row<4 * (CI + 1), QT, NBUF>(qk, d, t, head_idx, n_heads, qk_scale, cost,                              \
                                            reinterpret_cast<float (*)[BUF]>(raw + (size_t)wave * ANY_WAVE_FLOATS), lane); \
        break;                                                                                                        \
    }
    switch (cls) {
        WT_ROW_CASE(0, 8)
        WT_ROW_CASE(1, 4)
        WT_ROW_CASE(2, 2)
        default:
        WT_ROW_CASE(3, 2)
    }
#undef WT_ROW_CASE
}

// 64 columns x all T token rows per workgroup of 16 waves: wave w owns rows w, w+16, ... (<= 16 rows for
// T <= 256), all loaded before the first use so that one round trip to L2/HBM covers the whole column.
constexpr int CN_WAVES = 16;
constexpr int CN_ROWS = (WT_MAX_TOKENS + CN_WAVES - 1) / CN_WAVES;
__global__ __launch_bounds__(64 * CN_WAVES) void colnorm_kernel(float *__restrict__ cost,
                                                                const wt_seg_desc *__restrict__ segs,
                                                                unsigned *__restrict__ segmax, int unit0, int n_units,
                                                                int skip_small) {
    // XCD-aware: workgroups go to the 8 XCDs round-robin in linear order (id = x + gridDim.x * y -> XCD id % 8).  A row of
    // the cost matrix is F * 4 bytes, not a multiple of 128: the 256-byte row segments of neighbouring column blocks
    // share a cache line, so ALL column blocks of a unit are given to ONE XCD (unit u -> XCD u % 8) and meet in its L2
    // instead of each fetching the shared line from HBM.  The launch pads gridDim.y to a multiple of 8.
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int q = lin >> 3;
    const int col_block = q % (int)gridDim.x;
    const int local_unit = 8 * (q / (int)gridDim.x) + (lin & 7);
    if (local_unit >= n_units) return;      // block-uniform (padding of the grid)
    const int unit = unit0 + local_unit;
    const wt_seg_desc d = segs[unit];
    const int F = d.F, T = d.T;
    if (col_block * 64 >= F || (skip_small && wt_small_unit(T, F))) return;  // block-uniform
    __shared__ double ssq[CN_WAVES][64];
    __shared__ float smx[CN_WAVES][64];
    __shared__ float snorm[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = col_block * 64 + lane;
    const bool valid = f < F;
    const bool masked_col = d.pad_from > 0 && f >= d.pad_from;  // 0 = no mask, like the reference's `if max_duration:`
    float *base = cost + d.cost_offset + (valid ? f : 0);

    float v[CN_ROWS];
#pragma unroll
    for (int r = 0; r < CN_ROWS; ++r) {
        const int t = wave + r * CN_WAVES;
        v[r] = (valid && t < T) ? base[(int64_t)t * F] : 0.f;
    }
    double ss = 0.0;
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < CN_ROWS; ++r) {
        const int t = wave + r * CN_WAVES;
        ss += (double)v[r] * (double)v[r];
        if (!masked_col || t == T - 1) mx = fmaxf(mx, v[r]);
    }
    ssq[wave][lane] = ss;
    smx[wave][lane] = mx;
    __syncthreads();
    if (wave == 0) {
        double tot = 0.0;
        float m = 0.f;
#pragma unroll
        for (int w = 0; w < CN_WAVES; ++w) {
            tot += ssq[w][lane];
            m = fmaxf(m, smx[w][lane]);
        }
        const float norm = sqrtf((float)tot);
        snorm[lane] = norm;
        float r = valid ? m / norm : 0.f;  // max_t(w/norm) == max_t(w)/norm: IEEE division is monotone
        r = wave_max(r);
        // fire-and-forget (no return value, nobody waits for it).  Folding cost[0,0] = min into this kernel was
        // measured and dropped: a last-tile-done protocol needs either a release fence (writes the L2 back: +30 %
        // on the stage) or a returning CAS on the critical path (colnorm 25 -> 68 us); a 4 us kernel is cheaper.
        if (lane == 0) atomicMax(segmax + unit, __float_as_uint(r));
    }
    __syncthreads();
    const float norm = snorm[lane];
    if (valid) {
#pragma unroll
        for (int r = 0; r < CN_ROWS; ++r) {
            const int t = wave + r * CN_WAVES;
            if (t < T) base[(int64_t)t * F] = (masked_col && t < T - 1) ? 0.f : -(v[r] / norm);
        }
    }
}

// transcribe.py:1568  cost[0,0] = cost.min()
__global__ void fix00_kernel(float *__restrict__ cost, const wt_seg_desc *__restrict__ segs, const unsigned *__restrict__ segmax,
                             int n_seg, int skip_small) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_seg && !(skip_small && wt_small_unit(segs[s].T, segs[s].F))) cost[segs[s].cost_offset] = -__uint_as_float(segmax[s]);
}

// Units of one F class (C = 4*ceil(F/256) elements per lane) share a rowmean instantiation.  When the caller passes
// the units grouped by class (the host layer sorts them), each class is launched over ITS units only, with its own
// largest T and F; otherwise every class launch spans all units and foreign blocks return at once (correct, but a
// batch of many short units then launches mostly empty blocks: 75 us instead of 25 for 160 real-shape units).
struct ClassRange {
    int lo = 0, n = 0, maxT = 0, maxF = 0;
    bool any = false;
};
static bool class_ranges(const wt_seg_desc *segs_host, int n_seg, ClassRange (&cls)[7]) {
    bool grouped = true;
    int last = -1;
    for (int i = 0; i < n_seg; ++i) {
        const int c = (segs_host[i].F + 255) / 256 - 1;
        ClassRange &r = cls[c];
        if (!r.any) { r.any = true; r.lo = i; }
        else if (last != c) grouped = false;   // class c seen before, with another class in between
        r.n = i - r.lo + 1;
        if (segs_host[i].T > r.maxT) r.maxT = segs_host[i].T;
        if (segs_host[i].F > r.maxF) r.maxF = segs_host[i].F;
        last = c;
    }
    return grouped;
}

// Launch plan.  Every unit is served by the rowmean instantiation of ITS OWN F class (C = 4 * ceil(F / 256) elements per
// lane), so that its cost matrix does not depend on what else is in the batch -- with one exception that costs nothing in
// that respect: classes 0 and 1 (F <= 512, where the reference's per-segment units live) share ONE launch of the
// 8-elements-per-lane instantiation when both are present, because that instantiation reproduces the 4-element one's
// rows bit for bit on F <= 256 (wt_cost_core.h: the softmax denominator's summation tree).  A mixed batch is
// launch-bound, not work-bound: classes 2..6 each get their own launch only when present.
struct LaunchGroup {
    int ci = -1;            // instantiation index: C = 4 * (ci + 1)
    int lo = 0, n = 0;      // unit range when grouped
    int maxT = 0, maxF = 0, f_lo = 0, f_hi = 0;
};
constexpr int MAX_GROUPS = 6;
static int plan_groups(const ClassRange (&cls)[7], LaunchGroup (&g)[MAX_GROUPS], bool &contiguous) {
    auto fill = [&](LaunchGroup &out, int c0, int c1, int ci) {
        out = LaunchGroup();
        out.ci = ci;
        out.f_lo = c0 * 256;
        out.f_hi = (c1 + 1) * 256;
        int lo = 1 << 30, hi = 0, members = 0;
        for (int c = c0; c <= c1; ++c)
            if (cls[c].any) {
                if (cls[c].lo < lo) lo = cls[c].lo;
                if (cls[c].lo + cls[c].n > hi) hi = cls[c].lo + cls[c].n;
                members += cls[c].n;
                if (cls[c].maxT > out.maxT) out.maxT = cls[c].maxT;
                if (cls[c].maxF > out.maxF) out.maxF = cls[c].maxF;
            }
        out.lo = lo;
        out.n = hi - lo;
        if (out.n != members) contiguous = false;     // the group's classes are not adjacent in the unit order
    };
    int ng = 0;
    if (cls[0].any && cls[1].any) fill(g[ng++], 0, 1, 1);
    else if (cls[0].any) fill(g[ng++], 0, 0, 0);
    else if (cls[1].any) fill(g[ng++], 1, 1, 1);
    for (int c = 2; c < 7; ++c)
        if (cls[c].any) fill(g[ng++], c, c, c);
    return ng;
}

template <typename QT>
static int launch_rowmean(const QT *qk, const wt_seg_desc *segs_dev, int n_seg, const LaunchGroup *groups, int n_groups,
                          bool grouped, const int32_t *head_idx, int n_heads, float qk_scale, float *cost,
                          unsigned *segstate, hipStream_t st, int medfilt_width = 9) {
    for (int k = 0; k < n_groups; ++k) {
        const LaunchGroup &g = groups[k];
        const dim3 grid((g.maxT + 3) / 4, grouped ? g.n : n_seg);
        const int unit0 = grouped ? g.lo : 0;
        if (medfilt_width != 9) {
#define WT_LAUNCH_MEDW(CI)                                                                                            \
    case CI:                                                                                                          \
        hipLaunchKernelGGL((rowmean_medw_kernel<4 * (CI + 1), QT>), grid, dim3(256), 0, st, qk, segs_dev, head_idx, n_heads, \
                           qk_scale, cost, segstate, unit0, g.f_lo, g.f_hi, medfilt_width);                            \
        break;
            switch (g.ci) {
                WT_LAUNCH_MEDW(0) WT_LAUNCH_MEDW(1) WT_LAUNCH_MEDW(2) WT_LAUNCH_MEDW(3) WT_LAUNCH_MEDW(4) WT_LAUNCH_MEDW(5)
                WT_LAUNCH_MEDW(6)
            }
#undef WT_LAUNCH_MEDW
            continue;
        }
#define WT_LAUNCH_ROWMEAN(CI)                                                                                        \
    case CI:                                                                                                         \
        if (g.f_lo == CI * 256)                                                                                      \
            hipLaunchKernelGGL((rowmean_kernel<4 * (CI + 1), QT, true>), grid, dim3(256), 0, st, qk, segs_dev, head_idx, \
                               n_heads, qk_scale, cost, segstate, unit0, g.f_lo, g.f_hi);                             \
        else                                                                                                         \
            hipLaunchKernelGGL((rowmean_kernel<4 * (CI + 1), QT, false>), grid, dim3(256), 0, st, qk, segs_dev, head_idx, \
                               n_heads, qk_scale, cost, segstate, unit0, g.f_lo, g.f_hi);                             \
        break;
        switch (g.ci) {
            WT_LAUNCH_ROWMEAN(0)
            WT_LAUNCH_ROWMEAN(1)
            WT_LAUNCH_ROWMEAN(2)
            WT_LAUNCH_ROWMEAN(3)
            WT_LAUNCH_ROWMEAN(4)
            WT_LAUNCH_ROWMEAN(5)
            WT_LAUNCH_ROWMEAN(6)
        }
#undef WT_LAUNCH_ROWMEAN
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int cost_batch(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
               const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, bool skip_small,
               bool rows_per_class, hipStream_t st) {
    if (!qk || !segs_host || !segs_dev || !head_idx || !cost || n_seg < 0 || n_heads <= 0) {
        set_error("wt_cost_batch: null pointer or bad count");
        return WT_E_BADARG;
    }
    if (n_seg == 0) return WT_OK;
    if (medfilt_width < 1 || medfilt_width > 9 || (medfilt_width & 1) == 0) {
        set_error("wt_cost_batch: medfilt_width=%d unsupported (odd widths 1..9; the reference always uses 9)", medfilt_width);
        return WT_E_UNSUPPORTED;
    }
    if (!(qk_scale > 0.f)) {   // (the row maximum is taken before the scaling; the reference always passes 1.0)
        set_error("wt_cost_batch: qk_scale=%g unsupported (must be > 0; the reference always uses 1.0)", (double)qk_scale);
        return WT_E_UNSUPPORTED;
    }
    int maxF = 0;
    for (int i = 0; i < n_seg; ++i) {
        const wt_seg_desc &d = segs_host[i];
        if (d.T < 1 || d.T > WT_MAX_TOKENS || d.F < 1 || d.F > WT_MAX_FRAMES || d.start_token < 0) {   // (colnorm covers 256 rows)
            set_error("wt_cost_batch: unit %d has unsupported shape T=%d F=%d start=%d", i, d.T, d.F, d.start_token);
            return WT_E_UNSUPPORTED;
        }
        if (d.F > maxF) maxF = d.F;
    }
    unsigned *segstate = nullptr;  // per unit max |cost| bits: zeroed by rowmean, merged by colnorm, used by fix00
    int rc = scratch(st, (size_t)n_seg * sizeof(unsigned), (void **)&segstate);
    if (rc) return rc;
    ClassRange cls[7];
    bool grouped = class_ranges(segs_host, n_seg, cls);
    LaunchGroup groups[MAX_GROUPS];
    const int n_groups = plan_groups(cls, groups, grouped);
    const int skip = skip_small ? 1 : 0;
    // A small batch: one launch for all of its classes, eight heads fetched at once (rowmean_any_kernel).  "Small" = fewer
    // token rows than two workgroups per CU hold at once; the bits are those of the per-class launches either way.
    long long total_rows = 0;
    for (int i = 0; i < n_seg; ++i) total_rows += segs_host[i].T;
    if (!rows_per_class && medfilt_width == 9 && total_rows <= 8LL * 256 * 2 && n_seg <= ANY_MAX_UNITS &&
        (qk_dtype == WT_DTYPE_F32 || qk_dtype == WT_DTYPE_F16)) {
        WgTable table;
        unsigned n_wg = 0;
        for (int i = 0; i < n_seg; ++i) {
            table.first[i] = (unsigned short)n_wg;
            n_wg += (unsigned)(segs_host[i].T + 3) / 4;
        }
        table.first[n_seg] = (unsigned short)n_wg;     // (total_rows <= 4096: at most 4096 / 4 + n_seg workgroups)
        const int merge01 = (cls[0].any && cls[1].any) ? 1 : 0;
        if (qk_dtype == WT_DTYPE_F32)
            hipLaunchKernelGGL(rowmean_any_kernel<float>, dim3(n_wg), dim3(256), 0, st, (const float *)qk, segs_dev, head_idx, n_heads,
                               qk_scale, cost, segstate, merge01, n_seg, table);
        else
            hipLaunchKernelGGL(rowmean_any_kernel<__half>, dim3(n_wg), dim3(256), 0, st, (const __half *)qk, segs_dev, head_idx,
                               n_heads, qk_scale, cost, segstate, merge01, n_seg, table);
        WT_HIP(hipGetLastError());
        rc = WT_OK;
        LaunchGroup longer[MAX_GROUPS];
        int n_longer = 0;
        for (int k = 0; k < n_groups; ++k)
            if (groups[k].ci > ANY_MAX_CLASS) longer[n_longer++] = groups[k];
        if (n_longer) {
            if (qk_dtype == WT_DTYPE_F32)
                rc = launch_rowmean((const float *)qk, segs_dev, n_seg, longer, n_longer, grouped, head_idx, n_heads, qk_scale, cost,
                                    segstate, st);
            else
                rc = launch_rowmean((const __half *)qk, segs_dev, n_seg, longer, n_longer, grouped, head_idx, n_heads, qk_scale, cost,
                                    segstate, st);
        }
    } else
    // the groups are contiguous unit ranges when the classes are (sorted input); else fall back to full grids
    if (qk_dtype == WT_DTYPE_F32)
        rc = launch_rowmean((const float *)qk, segs_dev, n_seg, groups, n_groups, grouped, head_idx, n_heads, qk_scale, cost,
                            segstate, st, medfilt_width);
    else if (qk_dtype == WT_DTYPE_F16)
        rc = launch_rowmean((const __half *)qk, segs_dev, n_seg, groups, n_groups, grouped, head_idx, n_heads, qk_scale, cost,
                            segstate, st, medfilt_width);
    else {
        set_error("wt_cost_batch: qk_dtype=%d", qk_dtype);
        return WT_E_BADARG;
    }
    if (rc) return rc;
    // wt_align_batch: the column pass of the small units belongs to the fused tail kernel (wt_small.hip); the kernels
    // below skip them, and are not launched at all for a batch of small units only
    bool any_big = !skip_small;
    for (int i = 0; i < n_seg && !any_big; ++i) any_big = !wt_small_unit(segs_host[i].T, segs_host[i].F);
    if (!any_big) return WT_OK;
    if (grouped) {
        for (int k = 0; k < n_groups; ++k) {
            bool big_here = !skip_small;   // (a group whose units all went to the fused tail kernel needs no column pass)
            for (int i = groups[k].lo; i < groups[k].lo + groups[k].n && !big_here; ++i)
                big_here = !wt_small_unit(segs_host[i].T, segs_host[i].F);
            if (!big_here) continue;
            hipLaunchKernelGGL(colnorm_kernel, dim3((groups[k].maxF + 63) / 64, (groups[k].n + 7) & ~7), dim3(64 * CN_WAVES), 0, st,
                               cost, segs_dev, segstate, groups[k].lo, groups[k].n, skip);
        }
    } else {
        hipLaunchKernelGGL(colnorm_kernel, dim3((maxF + 63) / 64, (n_seg + 7) & ~7), dim3(64 * CN_WAVES), 0, st, cost, segs_dev,
                           segstate, 0, n_seg, skip);
    }
    hipLaunchKernelGGL(fix00_kernel, dim3((n_seg + 255) / 256), dim3(256), 0, st, cost, segs_dev, segstate, n_seg, skip);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
