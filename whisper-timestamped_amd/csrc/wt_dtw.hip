// Batched DTW (dtw-python "symmetric1" semantics) + in-kernel backtrack + jumps.
//
// Replaces /root/reference/whisper_timestamped/transcribe.py:1572,1581
//     alignment = dtw.dtw(weights, step_pattern=dtw.stepPattern.symmetric1)
// and :1648-1652 (jumps from alignment.index1s/index2s).  The arithmetic of
// dtw-python (dtw_core.c:computeCM + _backtrack.py) is reproduced operation by
// operation: f64, candidates in the order
//     p1 = g[i-1,j-1] + c   p2 = g[i,j-1] + c   p3 = g[i-1,j] + c
// compared as SUMS with strict '<' (first wins), out-of-range candidates never
// win.  Each cell therefore gets bit-identical g and direction, and the integer
// outputs (jumps, path) are bit-exact for a given cost matrix.
//
// Mapping: one workgroup per unit, one LANE PER TOKEN ROW (wave w owns rows
// 64w..64w+63).  A wave sweeps anti-diagonals: at local step s lane l is at
// frame j = s - l; g[i-1,*] arrives from lane l-1 through one DPP wave_shr:1
// (no LDS, no barrier).  Waves are pipelined, not barrier-stepped: the last
// row of wave w is streamed to LDS (bnd[w][j]) and published every 16 frames
// through an LDS progress word that wave w+1 polls, so the whole unit costs
// F + 64*W (+ ~16 per wave hop) dependent steps instead of a barrier per
// anti-diagonal.  Directions (2 bit) are packed 16 steps per word into LDS
// (<= 117 KB) and never touch HBM; each lane streams its own cost row with
// 16-frame register prefetch.  Algorithmic HBM bytes: T*F*4 read + 4*(T+1)
// written.  The dependency chain (F+T cells) makes one unit latency-bound;
// throughput comes from the batch.
#include "wt_common.h"

namespace wt {

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void load16(const float *__restrict__ row, int j0, int F, bool row_ok, float (&dst)[16]) {
    if (row_ok && j0 >= 0 && j0 + 16 <= F) {
        const float4u *p = reinterpret_cast<const float4u *>(row + j0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4u r = p[k];
            dst[4 * k] = r.x; dst[4 * k + 1] = r.y; dst[4 * k + 2] = r.z; dst[4 * k + 3] = r.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = j0 + k;
            dst[k] = (row_ok && j >= 0 && j < F) ? row[j] : 0.f;
        }
    }
}

__device__ __forceinline__ double readlane_f64(double v, int k) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], k);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], k);
    return u.d;
}

__host__ __device__ inline int dtw_pitch(int F) { return (((F + 63) + 15) >> 4) | 1; }

__global__ void dtw_kernel(const float *__restrict__ cost, const wt_seg_desc *__restrict__ segs, int32_t *__restrict__ jumps,
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

    uint32_t *dirs = reinterpret_cast<uint32_t *>(smem);                         // [nw*64][pitch]
    double *bnd = reinterpret_cast<double *>(smem + (size_t)nw * 64 * pitch * 4);  // [nw-1][F]
    int *prog = reinterpret_cast<int *>(bnd + (size_t)(nw - 1) * F);              // [nw-1]
    if (threadIdx.x < nw) prog[threadIdx.x] = 0;
    __syncthreads();

    const float *crow = cost + d.cost_offset + (int64_t)(row_ok ? i : 0) * F;
    const double INF = __builtin_inf();
    double g = INF;         // g[i, j-1]
    double diag = INF;      // g[i-1, j-1]
    double gfinal = 0.0;
    float cur[16], nxt[16];
    load16(crow, -lane, F, row_ok, cur);

    for (int s0 = 0; s0 < nsteps; s0 += 16) {
        load16(crow, s0 + 16 - lane, F, row_ok, nxt);  // prefetch next block (one block ~ 1k cycles ahead)

        double bvals = INF;
        if (wave > 0) {
            const int need = min(s0 + 16, F);
            while (__hip_atomic_load(&prog[wave - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
                __builtin_amdgcn_s_sleep(2);
            const int jb = s0 + lane;
            if (lane < 16 && jb < F) bvals = bnd[(size_t)(wave - 1) * F + jb];
        }

        uint32_t dw = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = s0 + k - lane;
            const double edge = (wave > 0) ? readlane_f64(bvals, k) : INF;  // g[64w-1, s0+k] for lane 0
            const double up = wave_shr1(g, edge);                          // g[i-1, j]
            const double c = (double)cur[k];
            const double p1 = diag + c;
            const double p2 = g + c;
            const double p3 = up + c;
            double best = p1;
            uint32_t dir = 1;
            if (p2 < best) { best = p2; dir = 2; }
            if (p3 < best) { best = p3; dir = 3; }
            if (k == 0 && s0 == 0 && i == 0) best = c;  // cm[0,0] = lm[0,0]
            diag = up;
            g = best;
            dw |= dir << (2 * k);
            if (j == F - 1) gfinal = best;
            if (wave < nw - 1 && lane == 63 && j >= 0 && j < F) bnd[(size_t)wave * F + j] = best;
        }
        dirs[(size_t)i * pitch + (s0 >> 4)] = dw;
        if (wave < nw - 1 && lane == 63) {
            const int done = min(max(s0 + 16 - 63, 0), F);  // frames of row 64w+63 finished so far
            __hip_atomic_store(&prog[wave], done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
    if (dist && i == T - 1) dist[blockIdx.x] = gfinal;
    __syncthreads();
    if (wave != 0) return;

    // ---- backtrack (dtw/_backtrack.py) + jumps (transcribe.py:1648-1652) ----
    int32_t *jp = jumps + d.jumps_offset;
    int bi = T - 1, bj = F - 1;
    int len = 1;
    if (lane == 0) jp[T] = F - 1;
    while (bi > 0) {
        const int li = bi & 63;
        const int s = bj + li;
        const int p = s & 15;
        const uint32_t word = dirs[(size_t)bi * pitch + (s >> 4)];
        // skip the run of "same token, previous frame" (dir 2) steps inside this word
        const uint32_t x = word ^ 0xAAAAAAAAu;
        uint32_t y = (x | (x >> 1)) & 0x55555555u;
        y &= (p == 15) ? 0xFFFFFFFFu : ((1u << (2 * p + 2)) - 1u);
        if (y == 0) {
            bj -= p + 1;
            len += p + 1;
            continue;
        }
        const int q = (31 - __builtin_clz(y)) >> 1;
        bj -= p - q;
        len += p - q;
        const uint32_t dir = (word >> (2 * q)) & 3u;
        if (lane == 0) jp[bi] = bj;
        if (dir == 1) { --bi; --bj; } else { --bi; }  // dir 3: previous token, same frame
        ++len;
    }
    len += bj;  // row 0: straight left to (0,0)
    if (lane == 0) {
        jp[0] = 0;
        if (path_len) path_len[blockIdx.x] = len;
    }
    if (path_i && path_j) {
        int32_t *pi = path_i + d.path_offset, *pj = path_j + d.path_offset;
        bi = T - 1; bj = F - 1;
        int pos = len - 1;
        while (true) {
            if (lane == 0) { pi[pos] = bi; pj[pos] = bj; }
            if (bi == 0 && bj == 0) break;
            uint32_t dir = 2;
            if (bi > 0) {
                const int s = bj + (bi & 63);
                dir = (dirs[(size_t)bi * pitch + (s >> 4)] >> (2 * (s & 15))) & 3u;
            }
            if (dir == 1) { --bi; --bj; } else if (dir == 2) { --bj; } else { --bi; }
            --pos;
        }
    }
}

size_t dtw_lds_bytes(int nw, int F) {
    return (size_t)nw * 64 * dtw_pitch(F) * 4 + (size_t)(nw - 1) * F * 8 + 16;
}

int dtw_batch(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, int32_t *jumps,
              int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, hipStream_t st) {
    if (!cost || !segs_host || !segs_dev || !jumps || n_seg < 0 || (!path_i != !path_j)) {
        set_error("wt_dtw_batch: null pointer or bad count");
        return WT_E_BADARG;
    }
    if (n_seg == 0) return WT_OK;
    int maxF[5] = {0, 0, 0, 0, 0};
    for (int s = 0; s < n_seg; ++s) {
        const wt_seg_desc &d = segs_host[s];
        if (d.T < 1 || d.T > WT_MAX_TOKENS || d.F < 1 || d.F > WT_MAX_FRAMES) {
            set_error("wt_dtw_batch: unit %d has unsupported shape T=%d F=%d", s, d.T, d.F);
            return WT_E_UNSUPPORTED;
        }
        const int nw = (d.T + 63) / 64;
        if (d.F > maxF[nw]) maxF[nw] = d.F;
    }
    static bool attr_set = false;
    if (!attr_set) {
        WT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
        attr_set = true;
    }
    for (int nw = 1; nw <= 4; ++nw) {
        if (maxF[nw] == 0) continue;
        const size_t lds = dtw_lds_bytes(nw, maxF[nw]);
        hipLaunchKernelGGL(dtw_kernel, dim3(n_seg), dim3(64 * nw), lds, st, cost, segs_dev, jumps, path_i, path_j, path_len,
                           dist);
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
