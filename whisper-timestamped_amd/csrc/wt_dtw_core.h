// Cell-level core of the DTW (dtw-python "symmetric1" semantics, see wt_dtw.hip), shared by the batched kernel
// (wt_dtw.hip: up to four pipelined waves per unit, cost streamed from HBM, direction planes in a scratch slot) and the
// fused small-unit kernel (wt_small.hip: one wave, cost and planes in LDS).
#pragma once
#include <type_traits>

#include "wt_common.h"

namespace wt {

constexpr int BLK = 32;  // steps per block = bits per direction word
__host__ __device__ inline int dtw_blocks(int F) { return (F + 63 + BLK - 1) / BLK; }        // 32-step blocks of a sweep
__host__ __device__ inline int dtw_bnd_pitch(int F) { return (F + 64 + BLK + 1) & ~1; }      // doubles per boundary row (even: 16-byte rows)
constexpr int DUMP = 64 + BLK;  // doubles per producer wave: where lanes 0..62 park the per-step store only lane 63 needs

// in-place wave_shr:1 -- lane 0 keeps what `up` already holds (its +inf)
__device__ __forceinline__ void shift_in(double &up, double g) {
    union { double d; int i[2]; } s, o;
    s.d = g;
    o.d = up;
    o.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xf, 0xf, false);
    o.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xf, 0xf, false);
    up = o.d;
}

// wa = 2wa + (a1 < a2), wb = 2wb + (b1 < b2): a compare into VCC and an add-with-carry per plane (hipcc emits
// cndmask + shift + or instead).  ONE asm statement for both planes: between two separate statements hipcc's hazard
// recogniser, blind to their contents, puts an s_nop that costs the wave a whole issue slot every step.
__device__ __forceinline__ void plane_bits(uint32_t &wa, uint32_t &wb, double a1, double a2, double b1, double b2) {
    asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                 "v_cmp_lt_f64 vcc, %4, %5\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc"
                 : "+v"(wa), "+v"(wb) : "v"(a1), "v"(a2), "v"(b1), "v"(b2) : "vcc");
}
// One 32-step block of the anti-diagonal sweep.
// EDGE: this wave has a producer wave above it; edge[k] = g[64w-1, s0+k] (the same value in every lane, read from
// the boundary row with broadcast LDS loads before the block) becomes the "old" operand of the wave_shr:1 that
// delivers g[i-1,*], i.e. what lane 0 receives -- no rotation, no copy: 2 DPP moves per step like the first wave.
// PUBLISH: a consumer wave below; every lane stores its `best` of step k at pub[k] -- lane 63's pointer walks the
// boundary row, the other lanes' pointers sit in a 64+32-double parking area (distinct addresses: no bank
// conflict, no exec juggling) -- one ds_write per step instead of a 4-DPP cross-lane shift register.
// u0/u1 alternate as "g[i-1,j]" and "g[i-1,j-1]" so that no register is copied.
// FIRST: the block that holds step 0 on the first wave.  Cell (0,0) is seeded through lane 0's `diag` (u1 = 0.0, so
// that p1 = 0 + lm[0,0] = cm[0,0] as in computeCM); wave_shr:1 never overwrites lane 0, so that seed must be
// retired to +inf before u1 comes back as "g[-1, 1]" at step 1 -- two moves, once per unit, none in the steady loop.
// NOUP: the reference's other step pattern (T.py:1575-1580, subwords_can_be_empty=False): "symmetric1 without the
// possibility to have the same timestamp for two tokens" = candidates p1 (diagonal) and p2 (same token, previous frame)
// only; the previous-token/same-frame candidate never exists, so its plane bit is never set.
template <bool EDGE, bool PUBLISH, bool DIST, bool FIRST = false, bool NOUP = false>
__device__ __forceinline__ void sweep_block(const float (&cur)[BLK], double &g, double &u0, double &u1,
                                            const double (&edge)[BLK], uint32_t &wa, uint32_t &wb, double *pub, int s0,
                                            int sfinal, double &gfinal) {
#pragma unroll
    for (int k = 0; k < BLK; ++k) {
        double &up = (k & 1) ? u1 : u0;          // g[i-1, j]   (written now)
        const double diag = (k & 1) ? u0 : u1;   // g[i-1, j-1] (written one step ago)
        if (FIRST && k == 1) u1 = __builtin_inf();   // (lanes > 0 receive their neighbour's g just below)
        if (EDGE) up = wave_shr1(g, edge[k]);    // lane 0 <- edge value of this step, lane l <- g of lane l-1
        else shift_in(up, g);                    // lane 0 keeps its +inf
        const double c = (double)cur[k];
        const double p1 = diag + c;
        const double p2 = g + c;
        const double p3 = NOUP ? __builtin_inf() : up + c;
        const double m12 = __builtin_fmin(p1, p2);
        const double best = __builtin_fmin(m12, p3);
        // plane A: "same token, previous frame" beats the diagonal; plane B: "previous token, same frame" beats both
        plane_bits(wa, wb, p2, p1, p3, m12);
        g = best;
        if (PUBLISH) pub[k] = best;
        if (DIST && s0 + k == sfinal) gfinal = best;
    }
}

// ---- backtrack (dtw/_backtrack.py) + jumps (transcribe.py:1648-1652), on ONE wave ----
// `rd(k)` returns the plane word pair k = block * rowsP + row (.x = plane A, .y = plane B).
template <typename RD>
__device__ __forceinline__ void backtrack_unit(RD rd, int T, int F, int rowsP, int lane, int unit_index, int32_t *jp,
                                               int32_t *pi, int32_t *pj, int32_t *path_len) {
    // step s of row r sits at bit (31 - (s & 31)) of word s >> 5:  A=1,B=0 -> dir 2; B=1 -> dir 3; else dir 1.
    // The walk is a chain of dependent steps on ONE wave, so what it must avoid is an LDS round trip (and taken
    // branches) per row.  Lane l loads the two plane words around the current step index for row bi - l (a 64-step
    // window: consecutive rows of a 64-row group move left by a few steps each, so one window serves several rows);
    // the walk itself runs on the scalar unit: v_readlane of that row's words, one 64-bit "first cell at or below
    // this step that is not direction 2" (s_ff1_i32_b64), and the row's jump goes into lane (row & 63) of one
    // VGPR; each 64-row group leaves with one coalesced store.  A new window is loaded only when the walk leaves it.
    int bi = T - 1;
    int r = bi & 63;
    int s = F - 1 + r;   // step index of the current cell = frame + (row & 63)
    int ups = 0;         // direction-3 moves: path length = F + ups
    int jv = 0;
    while (bi > 0) {
        // window: words win, win + 1 of rows bi, bi - 1, ... (lane l: row bi - l)
        const int win = max((s >> 5) - 1, 0);
        const int base = 32 * win;
        const int top = bi;
        const int rowp = win * rowsP + max(bi - lane, 0);                      // lanes read consecutive rows: coalesced
        const uint2 w0 = rd(rowp), w1 = rd(rowp + rowsP);
        const int a0 = (int)w0.x, b0 = (int)w0.y, a1 = (int)w1.x, b1 = (int)w1.y;
        bool more;
        do {   // one row per iteration, no memory access, one taken branch
            const int sel = top - bi;
            const uint64_t A = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(a0, sel) << 32) | (uint32_t)__builtin_amdgcn_readlane(a1, sel);
            const uint64_t B = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(b0, sel) << 32) | (uint32_t)__builtin_amdgcn_readlane(b1, sel);
            // cells at window positions <= s - base (bit 63 - position) that are NOT direction 2
            const uint64_t stop = (~A | B) & (~0ull << (base + 63 - s));
            if (__builtin_expect(stop == 0, 0)) {   // direction 2 down to the window's edge: go on in the words below
                s = base - 1;
                break;
            }
            const int bit = __builtin_ctzll(stop);
            s = base + 63 - bit;                              // the cell where the path leaves the row
            jv = (lane == r) ? s - r : jv;                    // jumps[bi] = its frame
            const int up = (int)(B >> bit) & 1;               // dir 3: previous token, same frame; dir 1: diagonal
            ups += up;
            s += up - 2;                                      // frame -= !up, row & 63 -= 1
            --r;
            --bi;
            more = ((s - base) | r | (bi - 1)) >= 0;      // still inside the window, the row group and the matrix
        } while (more);
        if (r < 0) {                                          // left a 64-row group: rows bi+1 .. bi+64
            if (bi + 1 + lane < T) jp[bi + 1 + lane] = jv;
            r = 63;
            s += 64;
        }
    }
    const int len = F + ups;
    if (lane > 0 && lane < T) jp[lane] = jv;
    if (lane == 0) {
        jp[0] = 0;
        jp[T] = F - 1;
        if (path_len) path_len[unit_index] = len;
    }
    if (pi && pj) {
        int bj = F - 1;
        bi = T - 1;
        int pos = len - 1;
        int have = -1;        // which (block, row) word pair is cached: the walk stays in a word for up to 32 steps
        uint32_t ca = 0, cb = 0;
        while (true) {
            if (lane == 0) { pi[pos] = bi; pj[pos] = bj; }
            if (bi == 0 && bj == 0) break;
            if (bi == 0) { --bj; --pos; continue; }
            const int s = bj + (bi & 63);
            const int key = (s >> 5) * rowsP + bi;
            if (key != have) {
                const uint2 AB = rd(key);
                ca = __builtin_amdgcn_readfirstlane(AB.x);
                cb = __builtin_amdgcn_readfirstlane(AB.y);
                have = key;
            }
            const int bit = 31 - (s & 31);
            const uint32_t a = (ca >> bit) & 1u;
            const uint32_t b = (cb >> bit) & 1u;
            if (b) { --bi; } else if (a) { --bj; } else { --bi; --bj; }
            --pos;
        }
    }
}

}  // namespace wt
