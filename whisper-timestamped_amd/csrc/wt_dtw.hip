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
// (no LDS, no barrier).  Waves are pipelined, not barrier-stepped: lane 63 of
// wave w streams its row to LDS (bnd[w][j], one ds_write per step) and
// publishes it every 32 frames through an LDS progress word that wave w+1
// polls; wave w+1 reads the 32 values of its next block with broadcast LDS
// loads and uses each one as the lane-0 operand of the wave_shr.  A unit costs
// about F + 64*W + 32*(W-1) dependent steps instead of a barrier per
// anti-diagonal, and a step is issue-bound (one wave = one instruction every
// ~4 cycles), so the per-step instruction count is what the design minimises.
// Direction bits (2 per cell, two planes) go to a per-unit slot of the library's
// scratch arena, one coalesced 8-byte store per lane per 32 steps ([block][row]
// layout: a wave writes 512 contiguous bytes; T*F/4 bytes per unit, written and
// read back by the same CU, i.e. through its XCD's L2) -- LDS holds only the
// boundary rows between waves (<= 47 KB), so several units share a CU and the
// kernels of another stream fit beside a unit (round 1 kept the planes in LDS:
// 141 KB per K-full unit, one unit per CU and nothing else next to it).  Each
// lane streams its own cost row with a 32-frame register prefetch.  Algorithmic
// HBM bytes: T*F*4 read + 4*(T+1) written.  One unit is latency-bound by its
// F+T-cell dependency chain; throughput comes from the batch.
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "wt_common.h"

namespace wt {

constexpr int BLK = 32;  // steps per block = bits per direction word

#ifdef WT_PROBE  // tools/probes/dtw_probe.hip only: per-wave timestamps (s_memtime) of unit 0
__device__ long long wt_probe_clk[16];
#define WT_STAMP(slot) do { if (blockIdx.x == 0 && lane == 0) wt_probe_clk[(slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define WT_STAMP(slot) do { } while (0)
#endif

// 32 consecutive cost values of this lane's row, starting at frame j0 (byte offset `boff` = 4 * (row*F + j0) into
// the unit).  ALWAYS exactly 8 dwordx4 loads, no branch: hipcc can then keep the prefetch in flight with a counted
// s_waitcnt (with a divergent slow path it fell back to vmcnt(0) at the first use of the previous block, i.e. the
// whole memory latency was exposed once per block: measured 3x on the kernel).  The loads are BUFFER loads on a
// descriptor of the unit (base = its first cost, range = its T*F floats + the 16 bytes of slack include/wtalign.h
// asks for): frames left of the row start read the previous row's tail (finite; those cells never feed a valid
// cell), offsets outside the unit -- negative ones wrap to huge unsigned values -- return 0 from the hardware range
// check.  One VALU instruction per block for the address instead of a clamp + 64-bit address per load.
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_blk(__amdgpu_buffer_rsrc_t unit, int boff, float (&dst)[BLK]) {
#pragma unroll
    for (int k = 0; k < BLK / 4; ++k) {
        const uint4v u = __builtin_amdgcn_raw_buffer_load_b128(unit, boff + 16 * k, 0, 0);
        const float4v r = __builtin_bit_cast(float4v, u);  // (whole vector: a bit_cast of ONE element reads element 0)
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
// The two plane words of a finished block -> the unit's scratch slot: ONE 8-byte buffer store per lane on a descriptor
// of the slot (base = the unit's first plane word, range = its slot): the compiler sees the store (it keeps its own
// vmcnt / hazard bookkeeping -- round 2 issued it as inline assembly behind the compiler's back) and the hardware
// range-checks it: a plane word can only land inside the unit's own slot, whatever the offset.  The backtrack reads
// the words back through the same descriptor (out-of-range reads return 0).
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_plane_words(__amdgpu_buffer_rsrc_t slot, int boff, uint32_t wa, uint32_t wb) {
    const uint2v data = {wa, wb};
    __builtin_amdgcn_raw_buffer_store_b64(data, slot, boff, 0, 0);
}
__device__ __forceinline__ uint2 load_plane_words(__amdgpu_buffer_rsrc_t slot, int boff) {
    const uint2v v = __builtin_amdgcn_raw_buffer_load_b64(slot, boff, 0, 0);
    return make_uint2(v.x, v.y);
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

template <bool DIST, bool TINY, bool NOUP>
__global__ __launch_bounds__(256) void dtw_kernel(const float *cost, const wt_seg_desc *__restrict__ segs, int32_t *__restrict__ jumps,
                           int32_t *__restrict__ path_i, int32_t *__restrict__ path_j, int32_t *__restrict__ path_len,
                           double *__restrict__ dist, uint2 *planes, long long plane_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const wt_seg_desc d = segs[blockIdx.x];
    const int T = d.T, F = d.F;
    const int nw = blockDim.x >> 6;
    if ((T + 63) / 64 != nw) return;  // block-uniform: unit belongs to another launch class

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: lives in an SGPR
    const int i = wave * 64 + lane;  // token row
    const bool row_ok = i < T;
    const int nsteps = F + 63;
    const int bpitch = dtw_bnd_pitch(F);
    const int rowsP = nw * 64;                               // rows of one block of plane words

    // [block][row] (.x = plane A word, .y = plane B word) in this unit's slot of the scratch arena
    uint2 *plane = planes + (size_t)blockIdx.x * plane_stride;
    const __amdgpu_buffer_rsrc_t pbuf = __builtin_amdgcn_make_buffer_rsrc(plane, 0, (int)plane_stride * 8, 0x00020000);
    double *bnd = reinterpret_cast<double *>(smem);          // [nw-1][bpitch], bnd[w][64 + j]
    double *park = bnd + (size_t)(nw - 1) * bpitch;          // [nw-1][DUMP]
    int *prog = reinterpret_cast<int *>(park + (size_t)(nw - 1) * DUMP);  // [nw-1]
    if (threadIdx.x < nw) prog[threadIdx.x] = 0;
    // The sweep is a dependency chain fed by ONE wave per SIMD: when another kernel's waves share the SIMD (a second
    // stream, the next step's kernels) every issue slot they take lengthens the chain.  Highest wave priority: the
    // arbiter serves this wave first and the others fill the slots it cannot use anyway.
    __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    WT_STAMP(wave);

    const float *unit = cost + d.cost_offset;
    if ((F < 4 || T * F < 36) != TINY) return;      // block-uniform: tiny units go to the element-wise instantiation
    const __amdgpu_buffer_rsrc_t ubuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(unit), 0, T * F * 4 + 16, 0x00020000);
    const int boff0 = 4 * ((row_ok ? i : T) * F - lane);  // byte offset of frame j = -lane of this lane's row
    const double INF = __builtin_inf();
    double g = INF;                      // g[i, j-1]
    double u0 = INF;                     // g[i-1, j] / g[i-1, j-1], alternating
    double u1 = (i == 0) ? 0.0 : INF;    // first diagonal: 0 + lm[0,0] reproduces cm[0,0] = lm[0,0]
    double gfinal = 0.0;
    const int sfinal = F - 1 + lane;
    uint32_t wa = 0, wb = 0;
    float bufA[BLK], bufB[BLK];
    if (TINY) load_blk_tiny(unit, i, -lane, T, F, bufA); else load_blk(ubuf, boff0, bufA);
    // where a producer wave stores its per-step `best`: lane 63 (row 64w+63, frame j = s - 63) -> bnd[w][64 + j] =
    // bnd[w][1 + s]; the other lanes -> their own slot of the parking area
    const bool producer = wave < nw - 1;
    const int pw = producer ? wave : 0;
    double *pub = (lane == 63) ? bnd + (size_t)pw * bpitch + 1 : park + (size_t)pw * DUMP + lane;
    const int pubinc = (lane == 63) ? BLK : 0;
    const double *erow = bnd + (size_t)(wave > 0 ? wave - 1 : 0) * bpitch;
    int pword = 8 * i;                                       // byte offset of this lane's word pair of the current block

    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    // The role of a wave (has a producer above / a consumer below) is decided ONCE, outside the sweep: each role runs
    // its own loop, so a block costs one taken branch instead of a chain of exec-masked skips over the other roles'
    // code (a taken branch is ~20 cycles for a lone wave, a skipped one ~11; measured ~15 cycles per step overall).
    auto sweep = [&](auto edge_c, auto publish_c) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_c)::value, PUBLISH = decltype(publish_c)::value;
        auto block = [&](const float (&cur)[BLK], float (&nxt)[BLK], int s0, auto first_c) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            if (TINY) load_blk_tiny(unit, i, s0 + BLK - lane, T, F, nxt);
            else {
                int vo = boff0 + 4 * (s0 + BLK);
                asm volatile("" : "+v"(vo));  // ONE address per block + immediate offsets (hipcc otherwise keeps 8 induction adds)
                load_blk(ubuf, vo, nxt);      // prefetch the next block (~2k cycles ahead)
            }
            // the loads stay HERE, a whole block ahead of their first use (without this fence hipcc rotates the loop
            // and sinks them to just before the use: the memory latency is then exposed once per iteration; `cost`
            // is deliberately not __restrict__: loads the compiler knows to be invariant ignore the fence)
            asm volatile("" ::: "memory");
            double edge[BLK];
            if (EDGE) {
                const int need = min(s0 + BLK, F);
                while (__hip_atomic_load(&prog[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                    __builtin_amdgcn_s_sleep(1);
                // frames >= F of the boundary row are never written: whatever is read there only feeds cells with
                // j >= F, which feed nothing valid; the block base is clamped so that the reads stay inside the row
                const double2 *e2 = reinterpret_cast<const double2 *>(erow + min(64 + s0, bpitch - BLK));
#pragma unroll
                for (int k = 0; k < BLK / 2; ++k) {
                    const double2 v = e2[k];
                    edge[2 * k] = v.x;
                    edge[2 * k + 1] = v.y;
                }
            }
            sweep_block<EDGE, PUBLISH, DIST, FIRST, NOUP>(cur, g, u0, u1, edge, wa, wb, pub, s0, sfinal, gfinal);
            store_plane_words(pbuf, pword, wa, wb);
            pword += 8 * rowsP;
            if (PUBLISH) {
                pub += pubinc;
                if (lane == 0) {
                    const int done = min(max(s0 + BLK - 63, 0), F);  // frames of row 64w+63 finished so far
                    __hip_atomic_store(&prog[wave], done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        };
        // (the first wave's first block is its own instantiation: see FIRST above; nsteps >= 64 = 2 blocks)
        block(bufA, bufB, 0, std::integral_constant<bool, !EDGE>{});
        block(bufB, bufA, BLK, no{});
        for (int s0 = 2 * BLK; s0 < nsteps; s0 += 2 * BLK) {
            block(bufA, bufB, s0, no{});
            if (s0 + BLK < nsteps) block(bufB, bufA, s0 + BLK, no{});
        }
    };
    if (wave == 0) {
        if (producer) sweep(no{}, yes{}); else sweep(no{}, no{});
    } else {
        if (producer) sweep(yes{}, yes{}); else sweep(yes{}, no{});
    }
    if (DIST && i == T - 1) dist[blockIdx.x] = gfinal;
    WT_STAMP(4 + wave);
    wait_vmcnt0();          // this wave's plane words have reached the L2 ...
    __syncthreads();        // ... before any wave of the workgroup reads them back
    if (wave != 0) return;
    WT_STAMP(8);

    // ---- backtrack (dtw/_backtrack.py) + jumps (transcribe.py:1648-1652) ----
    // step s of row r sits at bit (31 - (s & 31)) of word s >> 5:  A=1,B=0 -> dir 2; B=1 -> dir 3; else dir 1.
    // The walk is a chain of dependent steps on ONE wave, so what it must avoid is an LDS round trip (and taken
    // branches) per row.  Lane l loads the two plane words around the current step index for row bi - l (a 64-step
    // window: consecutive rows of a 64-row group move left by a few steps each, so one window serves several rows);
    // the walk itself runs on the scalar unit: v_readlane of that row's words, one 64-bit "first cell at or below
    // this step that is not direction 2" (s_ff1_i32_b64), and the row's jump goes into lane (row & 63) of one
    // VGPR; each 64-row group leaves with one coalesced store.  A new window is loaded only when the walk leaves it.
    int32_t *jp = jumps + d.jumps_offset;
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
        const int rowp = 8 * (win * rowsP + max(bi - lane, 0));                // lanes read consecutive rows: coalesced
        const uint2 w0 = load_plane_words(pbuf, rowp), w1 = load_plane_words(pbuf, rowp + 8 * rowsP);
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
        if (path_len) path_len[blockIdx.x] = len;
    }
    WT_STAMP(9);
    if (path_i && path_j) {
        int32_t *pi = path_i + d.path_offset, *pj = path_j + d.path_offset;
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
                const uint2 AB = load_plane_words(pbuf, 8 * key);
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

static size_t dtw_plane_words(int nw, int F) { return (size_t)nw * 64 * dtw_blocks(F); }   // uint2 per unit slot

size_t dtw_lds_bytes(int nw, int F) {   // boundary rows + parking areas + progress words
    return (size_t)(nw - 1) * (dtw_bnd_pitch(F) + DUMP) * 8 + 16;
}

int scratch_dtw(hipStream_t st, size_t bytes, void **out);   // the direction planes' arena of (device, stream)

template <bool DIST, bool TINY, bool NOUP>
static int launch_dtw(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int *maxF, int32_t *jumps,
                      int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, uint2 *planes, hipStream_t st) {
    // function attributes are per (instantiation, device): set the first time each device launches this one
    static std::mutex mu;
    static bool attr_set[64] = {false};
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        set_error("wt_dtw_batch: device ordinal %d out of range", dev);
        return WT_E_UNSUPPORTED;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!attr_set[dev]) {
            WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_kernel<DIST, TINY, NOUP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set[dev] = true;
        }
    }
    for (int nw = 1; nw <= 4; ++nw) {
        if (maxF[nw] == 0) continue;
        const size_t lds = dtw_lds_bytes(nw, maxF[nw]);
        // (the launches of one call run one after the other on the stream: they share the plane slots)
        hipLaunchKernelGGL((dtw_kernel<DIST, TINY, NOUP>), dim3(n_seg), dim3(64 * nw), lds, st, cost, segs_dev, jumps, path_i,
                           path_j, path_len, dist, planes, (long long)dtw_plane_words(nw, maxF[nw]));
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

template <bool NOUP>
static int launch_all(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int *maxF, const int *maxFt,
                      int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, uint2 *planes,
                      hipStream_t st) {
    int rc = dist ? launch_dtw<true, false, NOUP>(cost, segs_dev, n_seg, maxF, jumps, path_i, path_j, path_len, dist, planes, st)
                  : launch_dtw<false, false, NOUP>(cost, segs_dev, n_seg, maxF, jumps, path_i, path_j, path_len, dist, planes, st);
    if (rc) return rc;
    return dist ? launch_dtw<true, true, NOUP>(cost, segs_dev, n_seg, maxFt, jumps, path_i, path_j, path_len, dist, planes, st)
                : launch_dtw<false, true, NOUP>(cost, segs_dev, n_seg, maxFt, jumps, path_i, path_j, path_len, dist, planes, st);
}

int dtw_batch(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, int step_pattern,
              int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    if (!cost || !segs_host || !segs_dev || !jumps || n_seg < 0 || (!path_i != !path_j)) {
        set_error("wt_dtw_batch: null pointer or bad count");
        return WT_E_BADARG;
    }
    if (step_pattern != WT_STEP_SYMMETRIC1 && step_pattern != WT_STEP_NO_EMPTY_SUBWORDS) {
        set_error("wt_dtw_batch: step_pattern=%d", step_pattern);
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
        if (step_pattern == WT_STEP_NO_EMPTY_SUBWORDS && d.T > d.F) {   // every token takes a frame: dtw-python finds no path
            set_error("wt_dtw_batch: unit %d has T=%d > F=%d: no warping path without same-frame token moves", s, d.T, d.F);
            return WT_E_UNSUPPORTED;
        }
    }
    size_t slot = 0;   // uint2 words per unit slot: the largest launch class of this call
    for (int nw = 1; nw <= 4; ++nw) {
        if (maxF[nw]) slot = std::max(slot, dtw_plane_words(nw, maxF[nw]));
        if (maxFt[nw]) slot = std::max(slot, dtw_plane_words(nw, maxFt[nw]));
    }
    uint2 *planes = nullptr;
    int rc = scratch_dtw(st, (size_t)n_seg * slot * sizeof(uint2), (void **)&planes);
    if (rc) return rc;
    return step_pattern == WT_STEP_SYMMETRIC1
               ? launch_all<false>(cost, segs_dev, n_seg, maxF, maxFt, jumps, path_i, path_j, path_len, dist, planes, st)
               : launch_all<true>(cost, segs_dev, n_seg, maxF, maxFt, jumps, path_i, path_j, path_len, dist, planes, st);
}

}  // namespace wt
