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

#include "wt_dtw_core.h"
#include "wt_small.h"

namespace wt {

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

template <bool DIST, bool TINY, bool NOUP>
__global__ __launch_bounds__(256) void dtw_kernel(const float *cost, const wt_seg_desc *__restrict__ segs, int32_t *__restrict__ jumps,
                           int32_t *__restrict__ path_i, int32_t *__restrict__ path_j, int32_t *__restrict__ path_len,
                           double *__restrict__ dist, uint2 *planes, long long plane_stride, int skip_small) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const wt_seg_desc d = segs[blockIdx.x];
    const int T = d.T, F = d.F;
    const int nw = blockDim.x >> 6;
    if ((T + 63) / 64 != nw) return;  // block-uniform: unit belongs to another launch class
    if (skip_small && wt_small_unit(T, F)) return;   // (wt_align_batch: the fused small-unit kernel owns this unit)

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

    // ---- backtrack + jumps (wt_dtw_core.h) ----
    backtrack_unit([&](int k) { return load_plane_words(pbuf, 8 * k); }, T, F, rowsP, lane, (int)blockIdx.x,
                   jumps + d.jumps_offset, (path_i && path_j) ? path_i + d.path_offset : nullptr,
                   (path_i && path_j) ? path_j + d.path_offset : nullptr, path_len);
    WT_STAMP(9);
}

static size_t dtw_plane_words(int nw, int F) { return (size_t)nw * 64 * dtw_blocks(F); }   // uint2 per unit slot

size_t dtw_lds_bytes(int nw, int F) {   // boundary rows + parking areas + progress words
    return (size_t)(nw - 1) * (dtw_bnd_pitch(F) + DUMP) * 8 + 16;
}

int scratch_dtw(hipStream_t st, size_t bytes, void **out);   // the direction planes' arena of (device, stream)

template <bool DIST, bool TINY, bool NOUP>
static int launch_dtw(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int *maxF, int32_t *jumps,
                      int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, uint2 *planes, int skip_small,
                      hipStream_t st) {
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
                           path_j, path_len, dist, planes, (long long)dtw_plane_words(nw, maxF[nw]), skip_small);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

template <bool NOUP>
static int launch_all(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int *maxF, const int *maxFt,
                      int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, uint2 *planes,
                      int skip_small, hipStream_t st) {
    int rc = dist ? launch_dtw<true, false, NOUP>(cost, segs_dev, n_seg, maxF, jumps, path_i, path_j, path_len, dist, planes, skip_small, st)
                  : launch_dtw<false, false, NOUP>(cost, segs_dev, n_seg, maxF, jumps, path_i, path_j, path_len, dist, planes, skip_small, st);
    if (rc) return rc;
    return dist ? launch_dtw<true, true, NOUP>(cost, segs_dev, n_seg, maxFt, jumps, path_i, path_j, path_len, dist, planes, skip_small, st)
                : launch_dtw<false, true, NOUP>(cost, segs_dev, n_seg, maxFt, jumps, path_i, path_j, path_len, dist, planes, skip_small, st);
}

int dtw_batch(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, int step_pattern,
              int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, bool skip_small,
              hipStream_t st) {
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
        if (skip_small && wt_small_unit(d.T, d.F)) continue;   // (its kernels skip it too)
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
    if (slot == 0) return WT_OK;   // (every unit belongs to the fused small-unit kernel)
    uint2 *planes = nullptr;
    int rc = scratch_dtw(st, (size_t)n_seg * slot * sizeof(uint2), (void **)&planes);
    if (rc) return rc;
    return step_pattern == WT_STEP_SYMMETRIC1
               ? launch_all<false>(cost, segs_dev, n_seg, maxF, maxFt, jumps, path_i, path_j, path_len, dist, planes, skip_small ? 1 : 0, st)
               : launch_all<true>(cost, segs_dev, n_seg, maxF, maxFt, jumps, path_i, path_j, path_len, dist, planes, skip_small ? 1 : 0, st);
}

}  // namespace wt
