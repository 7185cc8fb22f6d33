// Log-mel front end for batches of 30 s chunks.
//
// Replaces openai-whisper audio.log_mel_spectrogram + pad_or_trim, which the
// reference calls at /root/reference/whisper_timestamped/transcribe.py:1213-1214
// (naive strategy, per segment crop) and reaches through model.transcribe()
// in the efficient strategy (constants mirrored at transcribe.py:44-47):
//   torch.stft(n_fft=400, hop=160, periodic hann, centre + reflect pad), drop
//   the last frame, |X|^2, mel filterbank (n_mels x 201), log10(clamp 1e-10),
//   max(x, max(x) - 8), (x + 4) / 4, zero-pad to n_frames.
//
// Not a GEMM: the 400-point real DFT is factored 20 x 20 (Cooley-Tukey, n =
// 20*n1 + n2, k = k1 + 20*k2).  Twenty lanes own one frame; each lane does a
// radix-20 butterfly entirely in registers; both stages pair n with 20-n and
// k with 10-k, and the 20th roots of unity are compile-time constants, so the
// zeros and +-1 fold away: 324 multiply-adds per lane instead of 1240, carried
// two per instruction (v_pk_*_f32).  The W400 twiddle comes from a 3.2 KB LDS
// table laid out [n2][k1], the transposition between the two stages goes
// through LDS.  Three frames per wave, twelve per 256-thread workgroup (a tile;
// 39 KB of LDS: 16 waves per CU), 250 tiles per 30 s chunk, walked by a
// persistent grid of four workgroups per CU that keeps its tables in LDS and
// prefetches the next tile's samples.  ~13 kFLOP per frame instead of 322 kFLOP
// for the direct DFT.  What bounds it (profiles/r2l_sq_counters.txt, DESIGN.md
// section 6): the VALU and LDS instruction streams of four waves per SIMD, about
// half of the kernel each; the LDS instruction forms are chosen by their gfx950
// cost (single ds_read_b64 on 64 banks, 32-bank dword reads on a padded span).
// Pass 1 writes log10(mel) and one maximum per tile (no atomics, nothing
// to reset); pass 2 reduces them per chunk, applies the clamp/scale and the
// zero padding.  Algorithmic bytes: 480000*4 read + n_mels*3000*4 written per
// chunk (the intermediate is re-read from L2).  The banded form of the
// filterbank is cached per stream and filterbank pointer.
#include <algorithm>
#include <cmath>
#include <mutex>

#include "wt_common.h"

namespace wt {

__constant__ float k_hann[400];

// 20th roots of unity as compile-time constants: after full unrolling every use below is a literal, so the
// zeros / ones / sign symmetries fold away at compile time.
__host__ __device__ constexpr float c20(int m) {
    m = ((m % 20) + 20) % 20;
    if (m > 10) m = 20 - m;
    return m == 0 ? 1.0f : m == 1 ? 0.9510565162951535f : m == 2 ? 0.8090169943749475f : m == 3 ? 0.5877852522924731f
         : m == 4 ? 0.30901699437494745f : m == 5 ? 0.0f : m == 6 ? -0.30901699437494745f : m == 7 ? -0.5877852522924731f
         : m == 8 ? -0.8090169943749475f : m == 9 ? -0.9510565162951535f : -1.0f;
}
__host__ __device__ constexpr float s20(int m) { return c20(m - 5); }  // sin(x) = cos(x - pi/2)
// acc += x * w with w a compile-time constant (0 and +-1 cost nothing / one add)
#define WT_MAC(acc, x, w)                         \
    do {                                          \
        constexpr float _w = (w);                 \
        if (_w == 1.0f) acc += (x);               \
        else if (_w == -1.0f) acc -= (x);         \
        else if (_w != 0.0f) acc = fmaf((x), _w, acc); \
    } while (0)
__constant__ float2 k_w400[400];  // [n2][k1] = (cos, sin)(2*pi*n2*k1/400): the twiddle of stage-2 input n2 for output row k1

constexpr int FPB = 12;             // frames per workgroup (a multiple of 4: the mel projection reads four frames per tap)
constexpr int SPAN = 160 * (FPB - 1) + 400;
// The span sits in LDS with 20 idle floats after every 160 samples: the three frames of a wave then start 180 floats
// apart (20 mod 32 banks), which makes the stage-1 reads (20 lanes per frame, 32-lane groups on 32 banks)
// conflict-free; unpadded, frames 0 and 2 of a wave (320 floats apart) sat on the same banks.
constexpr int SPAD = 20;
constexpr int SPAN_LDS = SPAN + SPAD * ((SPAN - 1) / 160);
constexpr int YP = 21;              // padded row of the stage-1 -> stage-2 exchange

__device__ __forceinline__ int enc_key(float v) {
    const int k = __float_as_int(v);
    return k >= 0 ? k : k ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec_key(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

constexpr int NNZ_CAP = 640;        // compact (banded) filterbank weights kept in LDS
constexpr int MAX_MELS = 256;

// ws layout (ints): lo[MAX_MELS] | n[MAX_MELS] | off[MAX_MELS] | total | weights[NNZ_CAP] | wgmax[n_chunks][n_wg] (float)
// Preparation, ONE workgroup of 16 waves, run only when the arena does not already hold the banded form of this
// filterbank (the result is cached per stream and (mel_fb pointer, n_mels)): dense (n_mels x 201) -> banded form.  A wave owns rows w, w+16, ...: all of its rows are fetched with coalesced loads
// issued back to back (one memory latency), first / last non-zero tap by wave reductions, 256-entry prefix sum in
// LDS, then the non-zero spans are copied.  ~4 us (a single-thread-per-row version took 48 us, an LDS-staged one 20).
constexpr int INIT_WAVES = 16;
constexpr int INIT_ROWS = MAX_MELS / INIT_WAVES;
__global__ __launch_bounds__(64 * INIT_WAVES) void logmel_init_kernel(int *ws, const float *__restrict__ fb, int n_mels) {
    __shared__ int s_lo[MAX_MELS], s_w[MAX_MELS], s_off[MAX_MELS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *lo = ws, *cnt = lo + MAX_MELS, *off = cnt + MAX_MELS, *total = off + MAX_MELS;
    float *wts = reinterpret_cast<float *>(total + 1);
    float v[INIT_ROWS][4];
#pragma unroll
    for (int r = 0; r < INIT_ROWS; ++r) {
        const int m = wave + r * INIT_WAVES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;
            v[r][j] = (m < n_mels && k < 201) ? fb[m * 201 + k] : 0.f;
        }
    }
#pragma unroll
    for (int r = 0; r < INIT_ROWS; ++r) {
        const int m = wave + r * INIT_WAVES;
        int first = 201, last = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v[r][j] != 0.f) {
                first = min(first, lane + 64 * j);
                last = max(last, lane + 64 * j + 1);
            }
        first = -wave_max_i(-first);
        last = wave_max_i(last);
        if (lane == 0 && m < MAX_MELS) {
            s_lo[m] = first < last ? first : 0;
            s_w[m] = first < last ? last - first : 0;
        }
    }
    __syncthreads();
    const int width = tid < MAX_MELS ? s_w[tid] : 0;
    if (tid < MAX_MELS) s_off[tid] = width;
    __syncthreads();
    for (int o = 1; o < MAX_MELS; o <<= 1) {  // inclusive Hillis-Steele scan over the 256 widths
        const int add = (tid < MAX_MELS && tid >= o) ? s_off[tid - o] : 0;
        __syncthreads();
        if (tid < MAX_MELS) s_off[tid] += add;
        __syncthreads();
    }
    const int tot = s_off[MAX_MELS - 1];
    if (tid < n_mels) { lo[tid] = s_lo[tid]; cnt[tid] = width; off[tid] = s_off[tid] - width; }
    if (tid == 0) *total = tot;
    if (tot <= NNZ_CAP)
        for (int e = tid; e < tot; e += 64 * INIT_WAVES) {  // element e of the concatenated bands: find its filter
            int a = 0, b = n_mels - 1;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (s_off[mid] > e) b = mid; else a = mid + 1;
            }
            const int k = e - (s_off[a] - s_w[a]);
            wts[e] = fb[a * 201 + s_lo[a] + k];
        }
}

// Persistent: the launch fills the chip once (4 workgroups per CU) and every workgroup walks tiles t, t + gridDim.x, ...
// (tile = 12 frames of one chunk, numbered chunk * n_wg + index).  The window, twiddle and filterbank tables go to LDS
// once per workgroup instead of once per tile, and the NEXT tile's PCM span is already travelling (three float4
// registers per thread, issued right after this tile's span has been published) while this tile is computed: a
// workgroup never sits through a global-memory latency with its LDS and registers idle.
struct TileInfo {
    int chunk, f0, nvs, nvf, i0;
    bool padding, interior;
};
__device__ __forceinline__ TileInfo tile_info(int t, int n_wg, const int32_t *__restrict__ n_valid_samples, int64_t n_samples,
                                              int n_frames, bool aligned) {
    TileInfo ti;
    ti.chunk = t / n_wg;
    ti.f0 = (t - ti.chunk * n_wg) * FPB;
    ti.nvs = n_valid_samples ? n_valid_samples[ti.chunk] : (int)n_samples;
    ti.nvf = min(ti.nvs / 160, n_frames);  // frames kept after dropping the last stft frame
    ti.padding = ti.f0 >= ti.nvf;          // whole tile is padding
    ti.i0 = ti.f0 * 160 - 200;             // centre=True: frame f covers padded[160f, 160f+400)
    // Interior tiles (no reflection at either end of the chunk) are a plain 16-byte-aligned copy: 540 float4 loads
    // for the workgroup; only the first / last tiles pay for the reflect index arithmetic.
    ti.interior = !ti.padding && aligned && ti.i0 >= 0 && ti.i0 + SPAN <= ti.nvs;
    return ti;
}

__global__ __launch_bounds__(256) void stft_mel_kernel(const float *__restrict__ pcm, int64_t n_samples,
                                                       const int32_t *__restrict__ n_valid_samples,
                                                       const float *__restrict__ fb, const int *__restrict__ ws,
                                                       float *__restrict__ wgmax, int n_mels, int n_frames,
                                                       float *__restrict__ mel_out, int n_wg, int n_tiles) {
    // 39.3 KB of LDS -> 4 workgroups (16 waves) per CU.  `pw` (stage-2 output) reuses the PCM span, which is dead
    // after stage 1 (a barrier separates them).
    static_assert(FPB * 204 >= SPAN_LDS && FPB * 204 >= 201 * FPB, "the span and the power spectrum share a buffer");
    static_assert(SPAD % 4 == 0 && 160 % 4 == 0, "a float4 of the span never straddles a padding gap");
    static_assert(SPAN % 4 == 0 && SPAN / 4 <= 3 * 256, "three float4 per thread carry a span");
    __shared__ __attribute__((aligned(16))) float span[FPB * 204];
    __shared__ float2 w400[400];
    __shared__ float2 yp[FPB][11][YP];   // stage-1 output, k1 = 0..10 (k1 > 10 is the conjugate of 20-k1)
    __shared__ float fbw[NNZ_CAP];
    __shared__ unsigned char fb_lo[MAX_MELS], fb_n[MAX_MELS];
    __shared__ unsigned short fb_off[MAX_MELS];
    __shared__ float hann[400];          // window taps: read below with immediate offsets (no per-lane address arithmetic)
    __shared__ float smax[4];
    float *pw = span;                    // power spectrum, [bin][frame of the tile]: four frames of a bin = one 16-byte read

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane / 20;       // frame slot inside the wave (0..2), lanes 60..63 idle
    const int u = lane - sub * 20;   // n2 in stage 1, k1 in stage 2
    const int slot = wave * 3 + sub;
    const bool aligned = (n_samples & 3) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0;

    // ---- tables: once per workgroup (published by the first tile's barrier) ----
    const int *g_lo = ws, *g_n = g_lo + MAX_MELS, *g_off = g_n + MAX_MELS, *g_tot = g_off + MAX_MELS;
    const float *g_w = reinterpret_cast<const float *>(g_tot + 1);
    const int nnz = *g_tot;
    const bool banded = nnz <= NNZ_CAP;
    for (int p = tid; p < 400; p += 256) hann[p] = k_hann[p];
    // twiddles laid out [n2][k1] (k_w400 is uploaded in that order): the 20 lanes of a frame read 20 consecutive float2
    // (W400^(n2*k1) indexed by its exponent put up to 10 lanes on one bank pair) at an immediate offset per n2
    for (int p = tid; p < 400; p += 256) w400[p] = k_w400[p];
    if (tid < n_mels) { fb_lo[tid] = (unsigned char)g_lo[tid]; fb_n[tid] = (unsigned char)g_n[tid]; fb_off[tid] = (unsigned short)g_off[tid]; }
    if (banded)
        for (int p = tid; p < nnz; p += 256) fbw[p] = g_w[p];

    // the span of an interior tile, in flight: float4 number tid, tid + 256, tid + 512 of its 540
    float4 pf0 = {0.f, 0.f, 0.f, 0.f}, pf1 = pf0, pf2 = pf0;
    auto prefetch = [&](const TileInfo &ti) {
        if (!ti.interior) return;                                // (block-uniform)
        const float4 *src = reinterpret_cast<const float4 *>(pcm + (int64_t)ti.chunk * n_samples + ti.i0);
        pf0 = src[tid];
        pf1 = src[tid + 256];
        if (tid + 512 < SPAN / 4) pf2 = src[tid + 512];
    };
    // XCD-aware walk: workgroups are dealt to the 8 XCDs round-robin (workgroup b -> XCD b % 8, each with its own L2), so
    // XCD x owns the CONTIGUOUS tile range [x N / 8, (x + 1) N / 8) and its workgroups take neighbouring tiles at the
    // same time: the 240 samples two neighbouring tiles share and the 48-byte pieces they write into the same
    // 128-byte lines of a mel row then meet in ONE L2 instead of travelling to HBM from two.
    const int XCDS = min(8, (int)gridDim.x);   // (fewer workgroups than XCDs: one range per workgroup)
    const int xcd = blockIdx.x % XCDS, rank_in_xcd = blockIdx.x / XCDS;
    const int wgs_in_xcd = ((int)gridDim.x - xcd + XCDS - 1) / XCDS;
    const int t_hi = (int)(((long long)(xcd + 1) * n_tiles) / XCDS);
    int t = (int)(((long long)xcd * n_tiles) / XCDS) + rank_in_xcd;
    TileInfo cur = {};
    if (t < t_hi) {
        cur = tile_info(t, n_wg, n_valid_samples, n_samples, n_frames, aligned);
        prefetch(cur);
    }
    for (; t < t_hi; t += wgs_in_xcd) {
    const int tn = t + wgs_in_xcd;
    TileInfo nxt = {};
    if (tn < t_hi) nxt = tile_info(tn, n_wg, n_valid_samples, n_samples, n_frames, aligned);
    const int chunk = cur.chunk, f0 = cur.f0, nvs = cur.nvs, nvf = cur.nvf;
    if (cur.padding) {                        // whole tile is padding (block-uniform)
        if (tid == 0) wgmax[t] = -INFINITY;
        prefetch(nxt);
        cur = nxt;
        continue;
    }
    const bool act = lane < 60 && (f0 + slot) < nvf;

    // ---- phase A: the tile's PCM span into LDS ----
    if (cur.interior) {
        float4 *dst = reinterpret_cast<float4 *>(span);
        constexpr int Q = SPAD / 4;               // float4 number p lands at p + Q * (p / 40)
        dst[tid + Q * (tid / 40)] = pf0;
        dst[tid + 256 + Q * ((tid + 256) / 40)] = pf1;
        if (tid + 512 < SPAN / 4) dst[tid + 512 + Q * ((tid + 512) / 40)] = pf2;
    } else {
        const float *x = pcm + (int64_t)chunk * n_samples;
        for (int p = tid; p < SPAN; p += 256) {
            int i = cur.i0 + p;
            if (i < 0) i = -i;           // reflect (no edge repeat)
            if (i >= nvs) i = 2 * (nvs - 1) - i;
            i = max(0, min(i, nvs - 1));
            span[p + SPAD * (p / 160)] = x[i];  // plain load: neighbouring tiles re-read 240 of these samples (non-temporal measured +20 %)
        }
    }
    __syncthreads();
    prefetch(nxt);                            // the next tile's span travels while this one is computed

    if (act) {
        // ---- stage 1: radix-20 over n1 for this lane's n2 = u (real input, k1 = 0..10) ----
        // Pair n1 with 20-n1 (cos even, sin odd) and k1 with 10-k1 ((-1)^n1 symmetry): 108 MACs instead of 440 -- and
        // carry them two per instruction (v_pk_*_f32): with e+[n] = a[n] + a[20-n], e-[n] = a[n] - a[20-n],
        //   Q_k = (Ao, -Be) = sum_j (e+[2j-1], e-[2j]) * ( cos((2j-1)k), -sin(2jk)) + (e+[9] cos(9k), 0)
        //   R_k = (-Bo, Ae) = sum_j (e-[2j-1], e+[2j]) * (-sin((2j-1)k),  cos(2jk)) + (-e-[9] sin(9k), 0)
        // and the outputs (re, -im) of rows k and 10-k are swap(R_k + (0, a0 +- a10)) +- Q_k.
        typedef float f2 __attribute__((ext_vector_type(2)));
        // The twenty samples and window taps are read as the PAIRS the packed arithmetic wants -- (a[2j-1], a[2j]),
        // (a[21-2j], a[20-2j]), (a0, a9), (a10, a11) -- by ds_read2_b32 with explicit offsets (left to itself hipcc pairs
        // neighbours (a[2m], a[2m+1]) and spends ~50 v_mov on re-pairing).  Offsets are in dwords (<= 255): sample n1
        // of a frame sits at 20 n1 + 20 (n1 / 8) in the padded span (SPAD), the window tap at 20 n1; second base at
        // sample 10 for n1 >= 12.  The compiler does not count these reads: one explicit wait below, tied to
        // the destination registers so that nothing that uses them is scheduled above it.
        typedef __attribute__((address_space(3))) const float lds_cf;
        const unsigned fr0 = (unsigned)(uintptr_t)(lds_cf *)(span + slot * (160 + SPAD) + u), fr1 = fr0 + 4 * (200 + SPAD);
        const unsigned hw0 = (unsigned)(uintptr_t)(lds_cf *)(hann + u), hw1 = hw0 + 800;
        f2 xa[10], xh[10];
#define WT_RD2(dst, base, o0, o1) asm volatile("ds_read2_b32 %0, %1 offset0:" #o0 " offset1:" #o1 : "=v"(dst) : "v"(base) : "memory")
        static_assert(SPAD == 20, "the immediate offsets below are written for SPAD = 20");
        WT_RD2(xa[0], fr0, 20, 40);   WT_RD2(xh[0], hw0, 20, 40);     // U1 = (a1, a2)
        WT_RD2(xa[1], fr1, 200, 180); WT_RD2(xh[1], hw1, 180, 160);   // V1 = (a19, a18)
        WT_RD2(xa[2], fr0, 60, 80);   WT_RD2(xh[2], hw0, 60, 80);     // U2 = (a3, a4)
        WT_RD2(xa[3], fr1, 160, 140); WT_RD2(xh[3], hw1, 140, 120);   // V2 = (a17, a16)
        WT_RD2(xa[4], fr0, 100, 120); WT_RD2(xh[4], hw0, 100, 120);   // U3 = (a5, a6)
        WT_RD2(xa[5], fr1, 100, 80);  WT_RD2(xh[5], hw1, 100, 80);    // V3 = (a15, a14)
        WT_RD2(xa[6], fr0, 140, 180); WT_RD2(xh[6], hw0, 140, 160);   // U4 = (a7, a8)
        WT_RD2(xa[7], fr1, 60, 40);   WT_RD2(xh[7], hw1, 60, 40);     // V4 = (a13, a12)
        WT_RD2(xa[8], fr0, 0, 200);   WT_RD2(xh[8], hw0, 0, 180);     // (a0, a9)
        WT_RD2(xa[9], fr0, 220, 240); WT_RD2(xh[9], hw0, 200, 220);   // (a10, a11)
#undef WT_RD2
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xa[4]), "+v"(xa[5]), "+v"(xa[6]), "+v"(xa[7]),
                       "+v"(xa[8]), "+v"(xa[9]), "+v"(xh[0]), "+v"(xh[1]), "+v"(xh[2]), "+v"(xh[3]), "+v"(xh[4]), "+v"(xh[5]),
                       "+v"(xh[6]), "+v"(xh[7]), "+v"(xh[8]), "+v"(xh[9]));
        f2 G[5], H[5];                          // G[j] = (e+[2j-1], e-[2j]),  H[j] = (e-[2j-1], e+[2j]),  j = 1..4
#pragma unroll
        for (int j = 1; j <= 4; ++j) {
            const f2 U = xa[2 * j - 2] * xh[2 * j - 2], V = xa[2 * j - 1] * xh[2 * j - 1];
            asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(G[j]) : "v"(U), "v"(V));   // (U.x + V.x, U.y - V.y)
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(H[j]) : "v"(U), "v"(V));   // (U.x - V.x, U.y + V.y)
        }
        const f2 X09 = xa[8] * xh[8], Y09 = xa[9] * xh[9];
        const f2 Se = X09 + Y09, So = X09 - Y09;   // (a0 + a10, e+[9]),  (a0 - a10, e-[9])
        const float ep9 = Se.y, em9 = So.y;
#define WT_FMA2(acc, x, wx, wy)                                                        \
    do {                                                                               \
        constexpr float _wx = (wx), _wy = (wy);                                        \
        if (_wx != 0.0f || _wy != 0.0f) acc = __builtin_elementwise_fma((x), (f2){_wx, _wy}, acc); \
    } while (0)
#define WT_S1(K)                                                                                                  \
        {                                                                                                         \
            f2 Q = {0.f, 0.f}, R = {0.f, 0.f};                                                                    \
            WT_FMA2(Q, G[1], c20(1 * K), -s20(2 * K)); WT_FMA2(Q, G[2], c20(3 * K), -s20(4 * K));                 \
            WT_FMA2(Q, G[3], c20(5 * K), -s20(6 * K)); WT_FMA2(Q, G[4], c20(7 * K), -s20(8 * K));                 \
            WT_MAC(Q.x, ep9, c20(9 * K));                                                                         \
            WT_FMA2(R, H[1], -s20(1 * K), c20(2 * K)); WT_FMA2(R, H[2], -s20(3 * K), c20(4 * K));                 \
            WT_FMA2(R, H[3], -s20(5 * K), c20(6 * K)); WT_FMA2(R, H[4], -s20(7 * K), c20(8 * K));                 \
            WT_MAC(R.x, em9, -s20(9 * K));                                                                        \
            R.y = ((K & 1) ? So.x : Se.x) + R.y;                                                                  \
            const f2 Rs = {R.y, R.x};                                                                             \
            const f2 lo = Rs + Q, hi = Rs - Q;                                                                    \
            yp[slot][K][u] = make_float2(lo.x, lo.y);                                                             \
            yp[slot][10 - K][u] = make_float2(hi.x, hi.y);                                                        \
        }
        WT_S1(0) WT_S1(1) WT_S1(2) WT_S1(3) WT_S1(4) WT_S1(5)
#undef WT_S1
#undef WT_FMA2
    }
    __syncthreads();
    if (act) {
        // ---- stage 2: twiddle by W400^(n2*k1), radix-20 over n2 for this lane's k1 = u ----
        // Rows k1 > 10 are the conjugates of rows 20 - k1 (real input).  Instead of conjugating twenty inputs, such a
        // lane multiplies by the CONJUGATE twiddle (the table holds it: upload_tables) -- that yields conj(b[n2]) --
        // and evaluates the radix-20 sums with conjugate roots (the sine terms change sign: `sg` below): the result is
        // conj(X[k]), whose power is the same.
        const int ks = u <= 10 ? u : 20 - u;
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 sg = u <= 10 ? (f2){1.f, -1.f} : (f2){-1.f, 1.f};
        f2 B[20];
        const float2 *wrow = w400 + u;
#pragma unroll
        for (int n2 = 0; n2 < 20; ++n2) {
            // volatile: keeps these forty reads as forty ds_read_b64 (2 LDS cycles each, 64 banks, conflict-free with
            // this layout); merged into ds_read2_b64 by the compiler they cost 8 cycles a pair on 32 banks
            typedef __attribute__((address_space(3))) const volatile f2 lds_vf2;
            const f2 v = *(lds_vf2 *)(&yp[slot][ks][n2]);
            const f2 w = *(lds_vf2 *)(&wrow[20 * n2]);  // (cos, +-sin) of W400^(n2*u)
            // (re + i im)(cos - i sin) = cos*(re, im) + sin*(im, -re): one packed multiply + one packed fma whose
            // operand modifiers do the swap and the sign (hipcc spends a v_xor and a v_mov on them)
            f2 b = v * (f2){w.x, w.x};
            asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(b) : "v"(v), "v"(w));
            B[n2] = b;
        }
        // X[k2] = sum_n2 b[n2] e^{-2 pi i n2 k2 / 20}: pair n2 with 20-n2 and k2 with 10-k2, and keep (re, im) as
        // one 64-bit operand so that each multiply-add is a v_pk_fma_f32: 108 packed MACs instead of 800 scalar ones.
        // The sine terms multiply by -i, i.e. swap (re, im) -> (im, -re): that swap is linear, so it is applied once
        // per output to the accumulated sum instead of once per input.
        f2 P[10], D[10];                       // P = b[n] + b[20-n],  D = b[n] - b[20-n]
#pragma unroll
        for (int n = 1; n < 10; ++n) {
            P[n] = B[n] + B[20 - n];
            D[n] = B[n] - B[20 - n];
        }
        const f2 Bev = B[0] + B[10], Bod = B[0] - B[10];
        float pk2[11];
#define WT_MAC2(acc, x, w)                                                          \
    do {                                                                            \
        constexpr float _w = (w);                                                   \
        if (_w == 1.0f) acc += (x);                                                 \
        else if (_w == -1.0f) acc -= (x);                                           \
        else if (_w != 0.0f) acc = __builtin_elementwise_fma((x), (f2){_w, _w}, acc); \
    } while (0)
#define WT_S2(K)                                                                                              \
        {                                                                                                     \
            f2 CDe = {0.f, 0.f}, CDo = {0.f, 0.f}, SDe = {0.f, 0.f}, SDo = {0.f, 0.f};                        \
            WT_MAC2(CDe, P[2], c20(2 * K)); WT_MAC2(CDe, P[4], c20(4 * K)); WT_MAC2(CDe, P[6], c20(6 * K));   \
            WT_MAC2(CDe, P[8], c20(8 * K));                                                                   \
            WT_MAC2(CDo, P[1], c20(1 * K)); WT_MAC2(CDo, P[3], c20(3 * K)); WT_MAC2(CDo, P[5], c20(5 * K));   \
            WT_MAC2(CDo, P[7], c20(7 * K)); WT_MAC2(CDo, P[9], c20(9 * K));                                   \
            WT_MAC2(SDe, D[2], s20(2 * K)); WT_MAC2(SDe, D[4], s20(4 * K)); WT_MAC2(SDe, D[6], s20(6 * K));   \
            WT_MAC2(SDe, D[8], s20(8 * K));                                                                   \
            WT_MAC2(SDo, D[1], s20(1 * K)); WT_MAC2(SDo, D[3], s20(3 * K)); WT_MAC2(SDo, D[5], s20(5 * K));   \
            WT_MAC2(SDo, D[7], s20(7 * K)); WT_MAC2(SDo, D[9], s20(9 * K));                                   \
            const f2 Bk = (K & 1) ? Bod : Bev;                                                                \
            const f2 U0 = Bk + (CDe + CDo), V0 = SDe + SDo, U1 = Bk + (CDe - CDo), V1 = SDo - SDe;            \
            /* X = U - i V (conjugate rows: U + i V): (re, im) = U + sg * (V.y, V.x), one packed fma */            \
            f2 X0, X1;                                                                                        \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(X0) : "v"(V0), "v"(sg), "v"(U0)); \
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(X1) : "v"(V1), "v"(sg), "v"(U1)); \
            /* |X|^2 directly: torch's stft.abs() ** 2 rounds through a square root, which moves the power by */ \
            /* <= 2 ulp (1e-7 of a log-mel value) and costs an IEEE sqrt per bin */                            \
            const f2 Q0 = X0 * X0, Q1 = X1 * X1;                                                              \
            pk2[K] = Q0.x + Q0.y; pk2[10 - K] = Q1.x + Q1.y;                                                  \
        }
        WT_S2(0) WT_S2(1) WT_S2(2) WT_S2(3) WT_S2(4) WT_S2(5)
#undef WT_S2
#undef WT_MAC2
#pragma unroll
        for (int k2 = 0; k2 < 10; ++k2) pw[(u + 20 * k2) * FPB + slot] = pk2[k2];
        if (u == 0) pw[200 * FPB + slot] = pk2[10];   // k = 200 (k1 = 0, k2 = 10)
    }
    __syncthreads();

    // ---- mel projection + log10; thread -> (mel m, group of 4 consecutive frames): every filter tap is read once
    //      and used for four frames (9 LDS/VALU operations per tap and 4 outputs instead of 24) ----
    float lmax = -INFINITY;
    constexpr int G = FPB / 4;
    for (int o = tid; o < n_mels * G; o += 256) {
        const int m = o / G, s0 = (o - m * G) * 4;
        if (f0 + s0 >= nvf) continue;
        const int lo = fb_lo[m], n = fb_n[m];
        typedef float f2 __attribute__((ext_vector_type(2)));
        const float4 *pp = reinterpret_cast<const float4 *>(pw + lo * FPB + s0);   // [tap k] at pp[3 * k] (FPB = 12 floats per bin)
        f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        if (banded) {
            const float *w = fbw + fb_off[m];
            for (int k = 0; k < n; ++k) {
                const float wk = w[k];
                const float4 p = pp[3 * k];
                a01 = __builtin_elementwise_fma((f2){p.x, p.y}, (f2){wk, wk}, a01);
                a23 = __builtin_elementwise_fma((f2){p.z, p.w}, (f2){wk, wk}, a23);
            }
        } else {                                   // more than NNZ_CAP taps: the weights stay in global memory
            const float *w = fb + m * 201 + lo;
            for (int k = 0; k < n; ++k) {
                const float wk = w[k];
                const float4 p = pp[3 * k];
                a01 = __builtin_elementwise_fma((f2){p.x, p.y}, (f2){wk, wk}, a01);
                a23 = __builtin_elementwise_fma((f2){p.z, p.w}, (f2){wk, wk}, a23);
            }
        }
        const float a0 = a01.x, a1 = a01.y, a2 = a23.x, a3 = a23.y;
        // log10 on v_log_f32 (log2, ~1 ulp) * log10(2): |error| < 3e-7 on values in [-10, 5]; ocml's log10f is 25 VALU
        const float L2 = 0.30102999566398120f;
        const float v0 = __builtin_amdgcn_logf(fmaxf(a0, 1e-10f)) * L2, v1 = __builtin_amdgcn_logf(fmaxf(a1, 1e-10f)) * L2;
        const float v2 = __builtin_amdgcn_logf(fmaxf(a2, 1e-10f)) * L2, v3 = __builtin_amdgcn_logf(fmaxf(a3, 1e-10f)) * L2;
        float *out = mel_out + ((int64_t)chunk * n_mels + m) * n_frames + f0 + s0;
        const int left = nvf - (f0 + s0);   // >= 1: frames of this group that exist
        out[0] = v0;
        lmax = fmaxf(lmax, v0);
        if (left > 1) { out[1] = v1; lmax = fmaxf(lmax, v1); }
        if (left > 2) { out[2] = v2; lmax = fmaxf(lmax, v2); }
        if (left > 3) { out[3] = v3; lmax = fmaxf(lmax, v3); }
    }
    lmax = wave_max(lmax);
    if (lane == 0) smax[wave] = lmax;
    __syncthreads();
    // per-tile maximum, one slot per tile: no atomics, nothing to reset between calls.  (This barrier also ends the
    // tile's reads of `pw`: the next iteration may overwrite the span.)
    if (tid == 0) wgmax[t] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    cur = nxt;
    }   // tiles
}

__global__ __launch_bounds__(256) void logmel_finalize_kernel(float *__restrict__ mel_out, const float *__restrict__ wgmax, int n_wg,
                                                              const int32_t *__restrict__ n_valid_samples, int64_t n_samples,
                                                              int n_mels, int n_frames, float *__restrict__ gmax) {
    const int chunk = blockIdx.y;

    const int nvs = n_valid_samples ? n_valid_samples[chunk] : (int)n_samples;
    const int nvf = min(nvs / 160, n_frames);
    __shared__ float s_mx[4];
    float m = -INFINITY;                      // the chunk's max of log10(mel): n_wg L2-resident floats per block
    for (int g = threadIdx.x; g < n_wg; g += 256) m = fmaxf(m, wgmax[(size_t)chunk * n_wg + g]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = m;
    __syncthreads();
    const float mx = fmaxf(fmaxf(s_mx[0], s_mx[1]), fmaxf(s_mx[2], s_mx[3]));
    const float floor_v = mx - 8.0f;
    float *base = mel_out + (int64_t)chunk * n_mels * n_frames;
    const int total = n_mels * n_frames;
    if ((n_frames & 3) == 0 && (reinterpret_cast<uintptr_t>(mel_out) & 15) == 0) {   // block-uniform: 16 bytes per thread
        float4 *base4 = reinterpret_cast<float4 *>(base);                             // (a row is a whole number of float4)
        for (int e4 = blockIdx.x * 256 + threadIdx.x; e4 < total / 4; e4 += gridDim.x * 256) {
            const int fr = (e4 * 4) % n_frames;
            float4 v = {0.f, 0.f, 0.f, 0.f};  // pad_or_trim: exact zeros
            if (fr < nvf) {
                const float4 x = base4[e4];
                v.x = (fmaxf(x.x, floor_v) + 4.0f) / 4.0f;
                if (fr + 1 < nvf) v.y = (fmaxf(x.y, floor_v) + 4.0f) / 4.0f;
                if (fr + 2 < nvf) v.z = (fmaxf(x.z, floor_v) + 4.0f) / 4.0f;
                if (fr + 3 < nvf) v.w = (fmaxf(x.w, floor_v) + 4.0f) / 4.0f;
            }
            base4[e4] = v;
        }
    } else {
        for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
            const int fr = e % n_frames;
            float v = 0.f;  // pad_or_trim: exact zeros
            if (fr < nvf) v = (fmaxf(base[e], floor_v) + 4.0f) / 4.0f;
            base[e] = v;
        }
    }
    if (gmax && blockIdx.x == 0 && threadIdx.x == 0) gmax[chunk] = mx;
}

// transcribe.py:1795-1805 find_start_padding of the windows logmel_finalize has just written, ONE WAVE per window.
// The reference walks back from column n_frames - 2 until a column differs from the (all-zero) last one.  Here the
// columns behind the valid frames are exact zeros BY CONSTRUCTION of the pass above, so the walk starts at the last
// valid column instead of re-reading up to 1500 columns of zeros (what wt_find_start_padding_batch has to do for a
// mel it knows nothing about: 18-65 us per batch); everything it decides, it decides on VALUES: the last column, and
// every column from the last valid one downwards until one is not all-zero (normally the first one it looks at).
// (Folding the decision into logmel_finalize itself was built first: one atomicMax per workgroup and window -- ~118
// device-scope atomics per address -- cost +35 us on the 11 us pass: profiles/r4d_bench.json, logmel stage 0.104 ms.)
__global__ __launch_bounds__(64) void padding_after_finalize_kernel(const float *__restrict__ mel, const int32_t *__restrict__ n_valid_samples,
                                                                    int64_t n_samples, int n_mels, int n_frames,
                                                                    int32_t *__restrict__ out) {
    const int chunk = blockIdx.x, lane = threadIdx.x;
    const float *base = mel + (int64_t)chunk * n_mels * n_frames;
    auto column_all_zero = [&](int c) {
        bool nz = false;
        for (int m = lane; m < n_mels; m += 64) nz |= base[(int64_t)m * n_frames + c] != 0.f;
        return __ballot(nz) == 0ull;                       // wave-uniform
    };
    if (!column_all_zero(n_frames - 1)) {                  // min == max == 0 fails -> None
        if (lane == 0) out[chunk] = -1;
        return;
    }
    const int nvs = n_valid_samples ? n_valid_samples[chunk] : (int)n_samples;
    const int nvf = min(nvs / 160, n_frames);
    int c = min(nvf, n_frames - 1) - 1;                    // columns nvf .. n_frames - 1 are zeros (written so above)
    while (c > 0 && column_all_zero(c)) --c;
    if (lane == 0) out[chunk] = c > 0 ? c + 1 : 0;
}

int scratch_tagged(hipStream_t st, size_t bytes, void **out, const void *tag_ptr, long long tag_val, bool *prepared);

static std::mutex g_tables_mu;
// CUs of the current device (cached per device ordinal; 256 on an MI355X)
static int device_cu_count() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lk(g_tables_mu);
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
// __constant__ symbols exist once PER DEVICE: the tables are uploaded the first time each device runs the front end.
static int upload_tables(hipStream_t st) {
    static bool done[64] = {false};
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) {
        set_error("wt_logmel_batch: device ordinal %d out of range", dev);
        return WT_E_UNSUPPORTED;
    }
    std::lock_guard<std::mutex> lk(g_tables_mu);
    static float hann[400];
    static float2 w400[400];
    if (done[dev]) return WT_OK;
    const double PI = 3.14159265358979323846;
    for (int n = 0; n < 400; ++n) {
        hann[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / 400.0));
        const int e = (n / 20) * (n % 20);   // n = 20*n2 + k1 -> exponent n2*k1 (<= 361)
        const double sgn = (n % 20) > 10 ? -1.0 : 1.0;   // rows k1 > 10 use the conjugate twiddle (see stage 2)
        w400[n] = make_float2((float)std::cos(2.0 * PI * e / 400.0), (float)(sgn * std::sin(2.0 * PI * e / 400.0)));
    }
    WT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(k_hann), hann, sizeof(hann), 0, hipMemcpyHostToDevice, st));
    WT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(k_w400), w400, sizeof(w400), 0, hipMemcpyHostToDevice, st));
    WT_HIP(hipStreamSynchronize(st));  // once per device
    done[dev] = true;
    return WT_OK;
}

int logmel_batch(const float *pcm, int n_chunks, int64_t n_samples, const int32_t *n_valid_samples, const float *mel_fb,
                 int n_mels, int n_frames, float *mel_out, float *gmax, int32_t *start_of_padding, hipStream_t st) {
    if (!pcm || !mel_fb || !mel_out || n_chunks < 0 || n_samples < 201 || n_mels <= 0 || n_mels > MAX_MELS || n_frames <= 0) {
        set_error("wt_logmel_batch: bad argument");
        return WT_E_BADARG;
    }
    if (n_chunks == 0) return WT_OK;
    int rc = upload_tables(st);
    if (rc) return rc;
    const int n_wg = (n_frames + FPB - 1) / FPB;
    const size_t head_ints = 3 * MAX_MELS + 1 + NNZ_CAP;
    int *ws = nullptr;
    bool prepared = false;
    rc = scratch_tagged(st, (head_ints + (size_t)n_chunks * n_wg) * sizeof(int), (void **)&ws, mel_fb, n_mels, &prepared);
    if (rc) return rc;
    float *wgmax = reinterpret_cast<float *>(ws + head_ints);
    // The banded filterbank is cached in the stream's arena: the contents behind `mel_fb` must not change while the
    // same pointer keeps being passed (wt_shutdown() or a different pointer / n_mels rebuilds it).
    if (!prepared) hipLaunchKernelGGL(logmel_init_kernel, dim3(1), dim3(64 * INIT_WAVES), 0, st, ws, mel_fb, n_mels);
    const long long n_tiles = (long long)n_wg * n_chunks;
    if (n_tiles > 0x7fffffffLL) {
        set_error("wt_logmel_batch: too many tiles");
        return WT_E_UNSUPPORTED;
    }
    const int resident = 4 * device_cu_count();   // 39.3 KB of LDS: four workgroups per CU
    hipLaunchKernelGGL(stft_mel_kernel, dim3((unsigned)std::min<long long>(n_tiles, resident)), dim3(256), 0, st, pcm, n_samples,
                       n_valid_samples, mel_fb, ws, wgmax, n_mels, n_frames, mel_out, n_wg, (int)n_tiles);
    const int total = n_mels * n_frames;
    int gx = (total + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3(gx, n_chunks), dim3(256), 0, st, mel_out, wgmax, n_wg, n_valid_samples,
                       n_samples, n_mels, n_frames, gmax);
    if (start_of_padding)
        hipLaunchKernelGGL(padding_after_finalize_kernel, dim3(n_chunks), dim3(64), 0, st, mel_out, n_valid_samples, n_samples,
                           n_mels, n_frames, start_of_padding);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
