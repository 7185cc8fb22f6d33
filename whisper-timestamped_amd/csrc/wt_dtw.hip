// Batched DTW (dtw-python "symmetric1" semantics) + in-kernel backtrack + jumps.
//
// Replaces /root/reference/whisper_timestamped/transcribe.py:1572,1581
//     alignment = dtw.dtw(weights, step_pattern=dtw.stepPattern.symmetric1)
// and :1648-1652 (jumps from alignment.index1s/index2s).  The arithmetic of
// dtw-python (dtw_core.c:computeCM + _backtrack.py) is reproduced operation by
// operation: f64, candidates in the order
//     p1 = g[i-1,j-1] + c   p2 = g[i,j-1] + c   p3 = g[i-1,j] + c
// compared as SUMS with strict '<' (first wins), out-of-range candidates never
// win.  The winning VALUE is min(min(p1,p2),p3) whatever the tie order, so the
// dependent chain per cell is add -> min -> min; the two strict compares that
// decide the DIRECTION (p2 < p1, p3 < min(p1,p2)) sit off that chain and are
// shifted into two per-lane bit planes by one add-with-carry each.  g and the
// directions are therefore bit-identical to dtw-python's, and the integer
// outputs (jumps, path) are bit-exact for a given cost matrix.
//
// Mapping: one workgroup per unit, one LANE PER TOKEN ROW (wave w owns rows
// 64w..64w+63).  A wave sweeps anti-diagonals: at local step s lane l is at
// frame j = s - l; g[i-1,*] arrives from lane l-1 through one DPP wave_shr:1
// (no LDS, no barrier).  Waves are pipelined, not barrier-stepped: the last
// row of wave w is streamed to LDS (bnd[w][j]) and published every 32 frames
// through an LDS progress word that wave w+1 polls, so a unit costs about
// F + 64*W + 32*(W-1) dependent steps instead of a barrier per anti-diagonal.
// Direction bits (2 per cell) live in LDS only (<= 117 KB); each lane streams
// its own cost row with a 32-frame register prefetch.  Algorithmic HBM bytes:
// T*F*4 read + 4*(T+1) written.  One unit is latency-bound by its F+T-cell
// dependency chain; throughput comes from the batch.
#include <mutex>

#include "wt_common.h"

namespace wt {

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int BLK = 32;  // steps per block = bits per direction word

#ifdef WT_PROBE  // tools/probes/dtw_probe.hip only: per-wave timestamps (s_memtime) of unit 0
__device__ long long wt_probe_clk[16];
#define WT_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0) wt_probe_clk[(slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define WT_STAMP(slot) do { } while (0)
#endif

// 32 consecutive cost values of this lane's row, starting at frame j0 (flat index `flat` = row*F + j0).
// ALWAYS exactly 8 dwordx4 loads, no branch: hipcc can then keep the prefetch in flight with a counted
// s_waitcnt (with a divergent slow path it fell back to vmcnt(0) at the first use of the previous block,
// i.e. the whole memory latency was exposed once per block: measured 3x on the kernel).  Frames outside
// [0,F) read neighbouring (finite) entries of the same unit -- those cells never feed a valid cell -- and
// the flat index is clamped into the unit; the last row may read up to 12 bytes past T*F (the read slack include/wtalign.h asks for).
__device__ __forceinline__ void load_blk(const float *__restrict__ unit, int flat, int last, float (&dst)[BLK]) {
#pragma unroll
    for (int k = 0; k < BLK / 4; ++k) {
        const int idx = min(max(flat + 4 * k, 0), last);
        const float4u r = *reinterpret_cast<const float4u *>(unit + idx);
        dst[4 * k] = r.x; dst[4 * k + 1] = r.y; dst[4 * k + 2] = r.z; dst[4 * k + 3] = r.w;
    }
}
// tiny units (F < 4 or T*F < 36): element-wise, still branch-free
__device__ __forceinline__ void load_blk_tiny(const float *__restrict__ unit, int row, int j0, int T, int F, float (&dst)[BLK]) {
#pragma unroll
    for (int k = 0; k < BLK; ++k) {
        const int j = j0 + k;
        const float v = unit[min(row, T - 1) * F + min(max(j, 0), F - 1)];
        dst[k] = (j >= 0 && j < F) ? v : 0.f;
    }
}

__host__ __device__ inline int dtw_pitch(int F) { return ((F + 63 + BLK - 1) / BLK) | 1; }  // words per row per plane
__host__ __device__ inline int dtw_bnd_pitch(int F) { return F + 64 + BLK; }                // doubles per boundary row

// in-place wave_shr:1 -- lane 0 keeps what `up` already holds (its +inf)
__device__ __forceinline__ void shift_in(double &up, double g) {
    union { double d; int i[2]; } s, o;
    s.d = g;
    o.d = up;
    o.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xf, 0xf, false);
    o.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xf, 0xf, false);
    up = o.d;
}

// Boundary feed of a consumer wave: lane k of `bv` holds g[64w-1, s0+k] (published by the wave above).  Each step
// the register is rotated down by one lane (DPP wave_shl:1), so lane 0 always holds the value of the current
// step and the wave_shr:1 that delivers g[i-1,*] takes it as its "old" operand -- no v_readlane -> SGPR -> VGPR
// round trip (measured +50 cycles per step), and the rotation sits off the dependent chain.
__device__ __forceinline__ void rotate_down(double &bv) {
    union { double d; int i[2]; } b;
    b.d = bv;
    b.i[0] = __builtin_amdgcn_update_dpp(b.i[0], b.i[0], 0x130, 0xf, 0xf, false);  // wave_shl:1, lane 63 keeps its own
    b.i[1] = __builtin_amdgcn_update_dpp(b.i[1], b.i[1], 0x130, 0xf, 0xf, false);
    bv = b.d;
}

// w = 2w + (a < b): one compare into VCC and one add-with-carry (hipcc emits cndmask + shift + or instead)
__device__ __forceinline__ void plane_bit(uint32_t &w, double a, double b) {
    asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(a), "v"(b) : "vcc");
}
// Cross-lane shift register for the boundary row: lane 0 <- best[lane 63], lane l <- acc[lane l-1].
// After 32 pushes lane l (< 32) holds the value of step 31-l.  All lanes execute it: no exec juggling, no LDS
// (a per-step ds_write from lane 63 measured +35 cycles/step as an all-lane same-address store and far more
// when predicated; these four DPP moves cost ~16).
__device__ __forceinline__ void push_lane63(double &acc, double best) {
    union { double d; int i[2]; } b, t, a;
    b.d = best;
    a.d = acc;
    t.i[0] = __builtin_amdgcn_mov_dpp(b.i[0], 0x13C, 0xf, 0xf, false);  // wave_ror:1 -> lane 0 = best[63]
    t.i[1] = __builtin_amdgcn_mov_dpp(b.i[1], 0x13C, 0xf, 0xf, false);
    t.i[0] = __builtin_amdgcn_update_dpp(t.i[0], a.i[0], 0x138, 0xf, 0xf, false);  // wave_shr:1, lane 0 keeps t
    t.i[1] = __builtin_amdgcn_update_dpp(t.i[1], a.i[1], 0x138, 0xf, 0xf, false);
    acc = t.d;
}

// One 32-step block of the anti-diagonal sweep.  EDGE: this wave has a
// producer wave above it (lane 0 takes g[i-1,*] from lane k of `bv`).
// u0/u1 alternate as "g[i-1,j]" and "g[i-1,j-1]" so that no register is copied.
template <bool EDGE, bool PUBLISH, bool DIST>
__device__ __forceinline__ void sweep_block(const float (&cur)[BLK], double &g, double &u0, double &u1, double bv,
                                            uint32_t &wa, uint32_t &wb, double &pubacc, int s0, int sfinal,
                                            double &gfinal) {
#pragma unroll
    for (int k = 0; k < BLK; ++k) {
        double &up = (k & 1) ? u1 : u0;          // g[i-1, j]   (written now)
        const double diag = (k & 1) ? u0 : u1;   // g[i-1, j-1] (written one step ago)
        if (EDGE) {
            up = wave_shr1(g, bv);  // lane 0 <- edge value of this step (lane 0 of bv), lane l <- g of lane l-1
            rotate_down(bv);
        } else {
            shift_in(up, g);        // lane 0 keeps its +inf
        }
        const double c = (double)cur[k];
        const double p1 = diag + c;
        const double p2 = g + c;
        const double p3 = up + c;
        const double m12 = __builtin_fmin(p1, p2);
        const double best = __builtin_fmin(m12, p3);
        plane_bit(wa, p2, p1);   // plane A: "same token, previous frame" beats the diagonal
        plane_bit(wb, p3, m12);  // plane B: "previous token, same frame" beats both
        g = best;
        if (PUBLISH) push_lane63(pubacc, best);
        if (DIST && s0 + k == sfinal) gfinal = best;
    }
}

template <bool DIST, bool TINY>
__global__ __launch_bounds__(256) void dtw_kernel(const float *__restrict__ cost, const wt_seg_desc *__restrict__ segs, int32_t *__restrict__ jumps,
                           int32_t *__restrict__ path_i, int32_t *__restrict__ path_j, int32_t *__restrict__ path_len,
                           double *__restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const wt_seg_desc d = segs[blockIdx.x];
    const int T = d.T, F = d.F;
    const int nw = blockDim.x >> 6;
    if ((T + 63) / 64 != nw) return;  // block-uniform: unit belongs to another launch class

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i = wave * 64 + lane;  // token row
    const bool row_ok = i < T;
    const int nsteps = F + 63;
    const int pitch = dtw_pitch(F);
    const int bpitch = dtw_bnd_pitch(F);

    uint2 *plane = reinterpret_cast<uint2 *>(smem);          // [nw*64][pitch] (.x = plane A word, .y = plane B word)
    double *bnd = reinterpret_cast<double *>(plane + (size_t)nw * 64 * pitch);  // [nw-1][bpitch], bnd[w][64 + j]
    double *dump = bnd + (size_t)(nw - 1) * bpitch;         // [BLK]
    int *prog = reinterpret_cast<int *>(dump + BLK);         // [nw-1]
    if (threadIdx.x < nw) prog[threadIdx.x] = 0;
    __syncthreads();
    WT_STAMP(wave);

    const float *unit = cost + d.cost_offset;
    if ((F < 4 || T * F < 36) != TINY) return;      // block-uniform: tiny units go to the element-wise instantiation
    const int last = T * F - 1;
    const int flat0 = (row_ok ? i : T) * F - lane;  // flat index of frame j = -lane of this lane's row
    const double INF = __builtin_inf();
    double g = INF;                      // g[i, j-1]
    double u0 = INF;                     // g[i-1, j] / g[i-1, j-1], alternating
    double u1 = (i == 0) ? 0.0 : INF;    // first diagonal: 0 + lm[0,0] reproduces cm[0,0] = lm[0,0]
    double gfinal = 0.0;
    const int sfinal = F - 1 + lane;
    uint32_t wa = 0, wb = 0;
    float bufA[BLK], bufB[BLK];
    if (TINY) load_blk_tiny(unit, i, -lane, T, F, bufA); else load_blk(unit, flat0, last, bufA);
    // where this lane publishes its g: lane 63 of a producer wave -> bnd[wave][64 + j], j = s - 63
    const bool producer = wave < nw - 1;
    double *pubrow = bnd + (size_t)(producer ? wave : 0) * bpitch + 1;  // step s of lane 63 -> pubrow[s] = bnd[w][64 + j]
    double pubacc = 0.0;

    auto block = [&](const float (&cur)[BLK], float (&nxt)[BLK], int s0) __attribute__((always_inline)) {
        if (TINY) load_blk_tiny(unit, i, s0 + BLK - lane, T, F, nxt);
        else load_blk(unit, flat0 + s0 + BLK, last, nxt);  // prefetch the next block (~2k cycles ahead)
        if (wave > 0) {
            const int need = min(s0 + BLK, F);
            while (__hip_atomic_load(&prog[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                __builtin_amdgcn_s_sleep(1);
            double bv = INF;
            if (lane < BLK && s0 + lane < F) bv = bnd[(size_t)(wave - 1) * bpitch + 64 + s0 + lane];
            if (producer) sweep_block<true, true, DIST>(cur, g, u0, u1, bv, wa, wb, pubacc, s0, sfinal, gfinal);
            else sweep_block<true, false, DIST>(cur, g, u0, u1, bv, wa, wb, pubacc, s0, sfinal, gfinal);
        } else {
            if (producer) sweep_block<false, true, DIST>(cur, g, u0, u1, INF, wa, wb, pubacc, s0, sfinal, gfinal);
            else sweep_block<false, false, DIST>(cur, g, u0, u1, INF, wa, wb, pubacc, s0, sfinal, gfinal);
        }
        plane[(size_t)i * pitch + s0 / BLK] = make_uint2(wa, wb);
        if (producer && lane < BLK) pubrow[s0 + BLK - 1 - lane] = pubacc;  // lane l holds step s0 + 31 - l
        if (producer && lane == 0) {
            const int done = min(max(s0 + BLK - 63, 0), F);  // frames of row 64w+63 finished so far
            __hip_atomic_store(&prog[wave], done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    for (int s0 = 0; s0 < nsteps; s0 += 2 * BLK) {
        block(bufA, bufB, s0);
        if (s0 + BLK < nsteps) block(bufB, bufA, s0 + BLK);
    }
    if (DIST && i == T - 1) dist[blockIdx.x] = gfinal;
    WT_STAMP(4 + wave);
    __syncthreads();
    if (wave != 0) return;
    WT_STAMP(8);

    // ---- backtrack (dtw/_backtrack.py) + jumps (transcribe.py:1648-1652) ----
    // step s of row r sits at bit (31 - (s & 31)) of word s >> 5:  A=1,B=0 -> dir 2; B=1 -> dir 3; else dir 1
    int32_t *jp = jumps + d.jumps_offset;
    int bi = T - 1, bj = F - 1;
    int len = 1;
    if (lane == 0) jp[T] = F - 1;
    while (bi > 0) {
        const int s = bj + (bi & 63);
        const int p = s & 31;
        const uint2 AB = plane[(size_t)bi * pitch + (s >> 5)];
        // wave-uniform walk: move it to the scalar unit (SALU ops issue in 1 cycle, no VALU dependent-issue latency)
        const uint32_t A = __builtin_amdgcn_readfirstlane(AB.x), B = __builtin_amdgcn_readfirstlane(AB.y);
        // run of dir-2 steps going down from step position p: bits (31-p) upward
        uint32_t notrun = ~(A & ~B);                    // 1 where dir != 2
        notrun &= 0xFFFFFFFFu << (31 - p);              // only positions <= p
        if (notrun == 0) {
            bj -= p + 1;
            len += p + 1;
            continue;
        }
        const int bit = __builtin_ctz(notrun);          // lowest set bit = highest step position <= p
        const int q = 31 - bit;
        bj -= p - q;
        len += p - q;
        if (lane == 0) jp[bi] = bj;
        if (!((B >> bit) & 1u)) --bj;                   // dir 1: diagonal; dir 3: previous token, same frame
        --bi;
        ++len;
    }
    len += bj;  // row 0: straight left to (0,0)
    if (lane == 0) {
        jp[0] = 0;
        if (path_len) path_len[blockIdx.x] = len;
    }
    WT_STAMP(9);
    if (path_i && path_j) {
        int32_t *pi = path_i + d.path_offset, *pj = path_j + d.path_offset;
        bi = T - 1; bj = F - 1;
        int pos = len - 1;
        while (true) {
            if (lane == 0) { pi[pos] = bi; pj[pos] = bj; }
            if (bi == 0 && bj == 0) break;
            if (bi == 0) { --bj; --pos; continue; }
            const int s = bj + (bi & 63);
            const uint2 AB = plane[(size_t)bi * pitch + (s >> 5)];
            const int bit = 31 - (s & 31);
            const uint32_t a = (__builtin_amdgcn_readfirstlane(AB.x) >> bit) & 1u;
            const uint32_t b = (__builtin_amdgcn_readfirstlane(AB.y) >> bit) & 1u;
            if (b) { --bi; } else if (a) { --bj; } else { --bi; --bj; }
            --pos;
        }
    }
}

size_t dtw_lds_bytes(int nw, int F) {
    return (size_t)2 * nw * 64 * dtw_pitch(F) * 4 + ((size_t)(nw - 1) * dtw_bnd_pitch(F) + BLK) * 8 + 16;
}

template <bool DIST, bool TINY>
static int launch_dtw(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int *maxF, int32_t *jumps,
                      int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    static std::once_flag once;  // per instantiation; function attributes are per process on one device
    hipError_t attr_rc = hipSuccess;
    std::call_once(once, [&] {
        attr_rc = hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_kernel<DIST, TINY>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    WT_HIP(attr_rc);
    for (int nw = 1; nw <= 4; ++nw) {
        if (maxF[nw] == 0) continue;
        const size_t lds = dtw_lds_bytes(nw, maxF[nw]);
        hipLaunchKernelGGL((dtw_kernel<DIST, TINY>), dim3(n_seg), dim3(64 * nw), lds, st, cost, segs_dev, jumps, path_i,
                           path_j, path_len, dist);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

int dtw_batch(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, int32_t *jumps,
              int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    if (!cost || !segs_host || !segs_dev || !jumps || n_seg < 0 || (!path_i != !path_j)) {
        set_error("wt_dtw_batch: null pointer or bad count");
        return WT_E_BADARG;
    }
    if (n_seg == 0) return WT_OK;
    int maxF[5] = {0, 0, 0, 0, 0}, maxFt[5] = {0, 0, 0, 0, 0};
    for (int s = 0; s < n_seg; ++s) {
        const wt_seg_desc &d = segs_host[s];
        if (d.T < 1 || d.T > WT_MAX_TOKENS || d.F < 1 || d.F > WT_MAX_FRAMES) {
            set_error("wt_dtw_batch: unit %d has unsupported shape T=%d F=%d", s, d.T, d.F);
            return WT_E_UNSUPPORTED;
        }
        const int nw = (d.T + 63) / 64;
        int *mf = (d.F < 4 || d.T * d.F < 36) ? maxFt : maxF;
        if (d.F > mf[nw]) mf[nw] = d.F;
    }
    if (dtw_lds_bytes(4, maxF[4] ? maxF[4] : 1) > 160 * 1024) {
        set_error("wt_dtw_batch: T>192 with F=%d needs more than 160 KiB of LDS", maxF[4]);
        return WT_E_UNSUPPORTED;
    }
    int rc = dist ? launch_dtw<true, false>(cost, segs_dev, n_seg, maxF, jumps, path_i, path_j, path_len, dist, st)
                  : launch_dtw<false, false>(cost, segs_dev, n_seg, maxF, jumps, path_i, path_j, path_len, dist, st);
    if (rc) return rc;
    return dist ? launch_dtw<true, true>(cost, segs_dev, n_seg, maxFt, jumps, path_i, path_j, path_len, dist, st)
                : launch_dtw<false, true>(cost, segs_dev, n_seg, maxFt, jumps, path_i, path_j, path_len, dist, st);
}

}  // namespace wt
