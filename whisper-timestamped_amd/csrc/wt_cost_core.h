// Row-level core of the local-cost construction, shared by the batched kernels (wt_cost.hip: one wave per token row,
// result to HBM) and the fused small-unit kernel (wt_small.hip: result stays in LDS for the DTW).  See wt_cost.hip for
// the reference lines this restates.
#pragma once
#include <hip/hip_fp16.h>

#include "wt_common.h"

namespace wt {

__device__ __forceinline__ float ld_qk(const float *p) { return *p; }
__device__ __forceinline__ float ld_qk(const __half *p) { return __half2float(*p); }

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// Asynchronous copy of one head's row (F logits) into dst[4 .. 4+F): HBM -> LDS directly (global_load_lds, no VGPR
// round trip, completion tracked by vmcnt).
// 16 bytes per lane and instruction (global_load_lds_dwordx4: lane l of a load lands at base + 16*l): a 1500-frame row
// is 6 instructions instead of 24.  Groups of four floats that do not lie fully inside the row re-read the last full
// group (their LDS slots are never used) -- except that the straddling group's slots hold the row's last F % 4
// elements: those come through a register (`tail`, lanes 0..2) and are written by the caller once the row has landed.
// Rows shorter than 4 frames take the dword form.  Returns this lane's tail element (meaningful for lane < F % 4).
template <int C>
__device__ __forceinline__ float stage_row(const float *__restrict__ src, float *dst, int F, int nch, int lane) {
    if (F < 4) {   // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + min(lane, F - 1)),
                                         (__attribute__((address_space(3))) void *)(dst + 4), 4, 0, 2);
        return 0.f;
    }
    const int nvec = F >> 2;
#pragma unroll
    for (int k = 0; k < C / 4; ++k) {
        if (k * 256 < F) {  // wave-uniform
            const int g = min(k * 64 + lane, nvec - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * g),
                                             (__attribute__((address_space(3))) void *)(dst + 4 + k * 256), 16, 0,
                                             2 /* cpol nt: every logit is read exactly once */);
        }
    }
    return src[min(4 * nvec + lane, F - 1)];
}
// fp16 rows (a build-side storage option, BASELINE config 5) take the same road: the halves travel HBM -> LDS AS THEY
// ARE (global_load_lds_dwordx4, 8 halves per lane and instruction: a 1500-frame row is 3 instructions) and are
// converted on the way from LDS to registers.  The DMA wants a 4-byte aligned source: a row that starts on an odd
// element is fetched from one element earlier (`sh` = 1: the same 4-byte word, i.e. inside the same tensor) and every
// LDS index below carries that shift.  hs[8 + sh + f] = element f; only full groups of 8 halves inside [0, F + sh) are
// copied, the last (F + sh) % 8 halves travel in a register (lanes 0..6) and are written once the row has landed,
// like the fp32 tail.  Returns this lane's tail half (raw bits).
template <int C>
__device__ __forceinline__ unsigned short stage_row_h(const __half *__restrict__ src, unsigned short *hs, int F, int sh, int lane) {
    const unsigned short *srcA = reinterpret_cast<const unsigned short *>(src) - sh;   // 4-byte aligned
    const int Fh = F + sh;
    const int nvec = Fh >> 3;
#pragma unroll
    for (int k = 0; k < (C + 7) / 8 + 1; ++k) {
        const int g = k * 64 + lane;
        if (g < nvec) {  // per lane: a lane without a full group neither reads nor writes (CAP is not a multiple of 512 halves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(srcA + 8 * g),
                                             (__attribute__((address_space(3))) void *)(hs + 8 + k * 512), 16, 0,
                                             2 /* cpol nt: every logit is read exactly once */);
        }
    }
    return srcA[min(8 * nvec + lane, Fh - 1)];
}

// The head mean of one token row, before the division: acc[q] (+)= softmax_F(median9(row of head a)) for every selected
// head a.  ONE WAVE; `lds` = this wave's two row buffers (double-buffered: head a+1 streams in while head a is being
// consumed); everything else lives in registers.  row0 = the token's row of flat head 0 at the window's first frame.
template <int C, typename QT>
struct RowBuf {
    static constexpr int CAP = C * 64;
    static constexpr bool HALF = sizeof(QT) == 2;
    // fp32: xs[4 + f] = element f (+ 4 halo slots each side).  fp16: raw halves, hs[8 + sh + f] = element f (sh = 0/1,
    // see stage_row_h), + halo, + the qword the shifted register load over-reads.
    static constexpr int BUF = HALF ? (CAP + 32) / 2 : CAP + 8;           // in floats (fp16: CAP + 32 halves)
};
typedef float f2 __attribute__((ext_vector_type(2)));

// NB = row buffers the wave owns.  NB = 2: head a+1 streams in while head a is being consumed (a launch with thousands of
// waves hides the rest of the latency).  NB > 2 (small launches, where a wave's fetches are what the kernel waits for): the
// rows of up to NB heads are fetched AT ONCE, one memory latency for all of them, then consumed one after the other.  The
// arithmetic and the order of accumulation over heads are the same: same bits.
// median of the MW (odd, <= 9) values centred on v[4] of the nine v[0..8]: transcribe.py:1546 with medfilt_width = MW
// (the reference's default and only caller value is 9; the other widths are the seam's parameter, transcribe.py:1439)
template <int MW>
__device__ __forceinline__ float median_window(const float *v);
template <>
__device__ __forceinline__ float median_window<1>(const float *v) { return v[4]; }
template <>
__device__ __forceinline__ float median_window<3>(const float *v) { return med3f(v[3], v[4], v[5]); }
template <>
__device__ __forceinline__ float median_window<5>(const float *v) {   // med3(e, max(min(a,b),min(c,d)), min(max(a,b),max(c,d)))
    const float a = v[2], b = v[3], c = v[5], d = v[6], e = v[4];
    return med3f(e, fmaxf(fminf(a, b), fminf(c, d)), fminf(fmaxf(a, b), fmaxf(c, d)));
}
template <>
__device__ __forceinline__ float median_window<7>(const float *v) {   // 13 compare-exchanges (Devillard's opt_med7)
    float p0 = v[1], p1 = v[2], p2 = v[3], p3 = v[4], p4 = v[5], p5 = v[6], p6 = v[7];
#define WT_CX(a, b) { const float lo_ = fminf(a, b), hi_ = fmaxf(a, b); a = lo_; b = hi_; }
    WT_CX(p0, p5) WT_CX(p0, p3) WT_CX(p1, p6) WT_CX(p2, p4) WT_CX(p0, p1) WT_CX(p3, p5) WT_CX(p2, p6)
    WT_CX(p2, p3) WT_CX(p3, p6) WT_CX(p4, p5) WT_CX(p1, p4) WT_CX(p1, p3) WT_CX(p3, p4)
#undef WT_CX
    return p3;
}

template <int C, typename QT, int NB = 2, int MW = 9>
__device__ __forceinline__ void head_sum_row(const QT *row0, int64_t head_stride, const int32_t *__restrict__ head_idx,
                                             int n_heads, int F, float qk_scale, float (*lds)[RowBuf<C, QT>::BUF], int lane,
                                             f2 (&acc)[C / 2]) {
    constexpr bool HALF = RowBuf<C, QT>::HALF;
    const int nch = (F + 63) >> 6;
    // halo duty of lanes 0..7: scipy 'reflect' source index of positions -4..-1 and F..F+3
    const int hpos = lane < 4 ? -(lane + 1) : F + (lane - 4);
    const int hsrc = reflect_index(hpos, F);

    // (re, re) pairs: the softmax tail below runs on packed fp32 instructions (v_pk_add/mul/fma_f32: two elements each)
#pragma unroll
    for (int q = 0; q < C / 2; ++q) acc[q] = (f2){0.f, 0.f};
    // exp((w - max) * qk_scale) = exp2((w - max) * qk_scale * log2(e)): one constant (the host refuses qk_scale <= 0).
    // The product is rounded once: half an ulp of the exponent, i.e. <= 1 ulp of the result for |w - max| < 2.9 and
    // growing only where exp() itself vanishes from the softmax sum (measured against the compensated two-term
    // product this kernel used before: same worst error against the oracle, 2.9e-7 of the matrix maximum; -7 % time)
    const f2 cexp = (f2){qk_scale * 1.44269502162933349609375f, qk_scale * 1.44269502162933349609375f};

    // fp32 rows: 16-byte LDS-DMA + a register for the last F % 4 elements; fp16 rows: the same with 8 halves per lane
    const int tail0 = F & ~3, ntail = (!HALF && F >= 4) ? (F & 3) : 0;
    // fp16: every head's row of this token starts at the same parity (head_stride elements apart: the shift is
    // recomputed per head, it is one AND)
    constexpr int NT = NB > 2 ? NB : 1;      // (the pipelined form has one row in flight: one set of tail registers)
    float tail[NT];
    unsigned short tailh[NT];
    int shv[NT];
    auto stage = [&](int a, int buf, int slot) __attribute__((always_inline)) {
        const QT *src = row0 + (int64_t)head_idx[a] * head_stride;
        if constexpr (HALF) {
            shv[slot] = (int)((reinterpret_cast<uintptr_t>(src) >> 1) & 1);
            tailh[slot] = stage_row_h<C>(src, reinterpret_cast<unsigned short *>(lds[buf]), F, shv[slot], lane);
        } else {
            tail[slot] = stage_row<C>(src, lds[buf], F, nch, lane);
        }
    };
    // the landed row of buffer `buf` -> registers x[0 .. C+8) (this lane's C elements + 4 of halo on each side)
    auto fetch = [&](int buf, int slot, float (&x)[C + 8]) __attribute__((always_inline)) {
        wave_lds_fence();
        if constexpr (HALF) {
            const int sh = shv[slot];
            unsigned short *hs = reinterpret_cast<unsigned short *>(lds[buf]);
            const int Fh = F + sh;
            if (lane < (Fh & 7)) hs[8 + (Fh & ~7) + lane] = tailh[slot];   // (the group that straddles the end of the row)
            wave_lds_fence();
            if (lane < 8) {
                const unsigned short hv = hs[8 + sh + hsrc];
                hs[8 + sh + hpos] = hv;
            }
            wave_lds_fence();
            // this lane's C + 8 elements start at half 4 + sh + lane * C: read the dwords from half 4 + lane * C (8-byte
            // aligned) and, for odd starts, funnel-shift neighbouring dwords by one half
            constexpr int NQ = (C + 8) / 4 + 1;
            const uint2 *qp = reinterpret_cast<const uint2 *>(hs + 4 + lane * C);
            unsigned w[2 * NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const uint2 r = qp[k];
                w[2 * k] = r.x; w[2 * k + 1] = r.y;
            }
            // wave-uniform, and a BRANCH: a row that starts on an even element (every row of a (.., T, 1500) ring does)
            // skips the funnel shift altogether -- as a select it was 32 of the 458 VALU instructions of a 24-wide head-row
            // (profiles/r4q_sq_counters_largev3_fp16.txt: the kernel is VALU-bound)
            if (__builtin_amdgcn_readfirstlane(sh) != 0) {
#pragma unroll
                for (int k = 0; k < (C + 8) / 2; ++k) {
                    const unsigned v = __builtin_amdgcn_alignbit(w[k + 1], w[k], 16);
                    const float2 f2v = __half22float2(*reinterpret_cast<const __half2 *>(&v));
                    x[2 * k] = f2v.x; x[2 * k + 1] = f2v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < (C + 8) / 2; ++k) {
                    const float2 f2v = __half22float2(*reinterpret_cast<const __half2 *>(&w[k]));
                    x[2 * k] = f2v.x; x[2 * k + 1] = f2v.y;
                }
            }
        } else {
            float *xs = lds[buf];  // xs[4+f] = element f; xs[0..3], xs[4+F..7+F] = reflected halo
            if (lane < ntail) xs[4 + tail0 + lane] = tail[slot];   // (the group that straddles the end of the row)
            wave_lds_fence();
            if (lane < 8) {
                const float hv = xs[4 + hsrc];
                xs[4 + hpos] = hv;
            }
            wave_lds_fence();
            const float4 *xp = reinterpret_cast<const float4 *>(xs + lane * C);
#pragma unroll
            for (int k = 0; k < (C + 8) / 4; ++k) {
                const float4 r = xp[k];
                x[4 * k + 0] = r.x; x[4 * k + 1] = r.y; x[4 * k + 2] = r.z; x[4 * k + 3] = r.w;
            }
        }
    };
    // median-9, softmax over the window, acc += softmax
    auto fold = [&](const float (&x)[C + 8]) __attribute__((always_inline)) {
        // median of 9 = med3(max3(lows), med3(mids), min3(highs)) over the sorted triples of 3 consecutive triples
        float lo[C + 6], mi[C + 6], hi[C + 6];
        if constexpr (MW == 9) {
#pragma unroll
            for (int p = 0; p < C + 6; ++p) {
                lo[p] = min3f(x[p], x[p + 1], x[p + 2]);
                mi[p] = med3f(x[p], x[p + 1], x[p + 2]);
                hi[p] = max3f(x[p], x[p + 1], x[p + 2]);
            }
        }
        float m[C];
        float mx = -1e30f;
#pragma unroll
        for (int q = 0; q < C; ++q) {
            float med;
            if constexpr (MW == 9)
                med = med3f(max3f(lo[q], lo[q + 3], lo[q + 6]), med3f(mi[q], mi[q + 3], mi[q + 6]),
                            min3f(hi[q], hi[q + 3], hi[q + 6]));
            else
                med = median_window<MW>(&x[q]);            // (x[q + 4] is element q of this lane)
            const bool ok = (lane * C + q) < F;
            m[q] = ok ? med : -1e30f;
            mx = fmaxf(mx, m[q]);
        }
        mx = wave_max_dpp(mx);
        const f2 mx2 = (f2){mx, mx};
        f2 e[C / 2];
#pragma unroll
        for (int q = 0; q < C / 2; ++q) {
            const f2 t = ((f2){m[2 * q], m[2 * q + 1]} - mx2) * cexp;
            e[q] = (f2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};   // (masked: exp2(-1.4e30) = 0)
        }
        // The softmax denominator is summed QUAD by quad of four consecutive frames, (e0 + e2) + (e1 + e3), a lane's quads
        // in order, then the wave tree of wave_sum_dpp.  For a row of F <= 256 frames this is the SAME binary tree in the
        // 4- and in the 8-elements-per-lane instantiation (quad index bits 0, 1, ... 5 in that order; lanes / quads beyond
        // the row contribute exact zeros), so a short unit's row does not depend on whether its batch also holds a
        // (256, 512] unit -- the one case in which units of two F classes share a launch (plan_groups).
        float lane_sum = 0.f;
#pragma unroll
        for (int k = 0; k < C / 4; ++k) {
            const f2 p = e[2 * k] + e[2 * k + 1];
            const float qs = p.x + p.y;
            lane_sum = k == 0 ? qs : lane_sum + qs;
        }
        const float s = wave_sum_dpp(lane_sum);
        const float inv = 1.0f / s;
        const f2 inv2 = (f2){inv, inv};
#pragma unroll
        for (int q = 0; q < C / 2; ++q) acc[q] = __builtin_elementwise_fma(e[q], inv2, acc[q]);
    };

    if constexpr (NB == 2) {
        stage(0, 0, 0);
        for (int a = 0; a < n_heads; ++a) {
            float x[C + 8];
            wait_vmcnt0();                 // head a's row has landed in LDS
            fetch(a & 1, 0, x);
            // Row a now lives in registers: start streaming head a+1 into the other buffer; it lands while
            // the VALU work below runs.  (Issued AFTER the LDS reads: hipcc drains vmcnt before any ds_read
            // that follows an LDS-DMA, which would serialise the copy with the reads.)
            if (a + 1 < n_heads) stage(a + 1, (a + 1) & 1, 0);
            fold(x);
        }
    } else {
        for (int a0 = 0; a0 < n_heads; a0 += NB) {
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (a0 + k < n_heads) stage(a0 + k, k, k);   // (wave-uniform)
            wait_vmcnt0();                 // all of them have landed
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (a0 + k < n_heads) {
                    float x[C + 8];
                    fetch(k, k, x);
                    fold(x);
                }
        }
    }
}



// mean over heads (torch CPU: sum then div): x * (1/n) == x / n exactly for a power of two
__device__ __forceinline__ float head_mean(float sum, int n_heads) {
    const float nh = (float)n_heads;
    return ((n_heads & (n_heads - 1)) == 0) ? sum * (1.0f / nh) : sum / nh;
}

}  // namespace wt
