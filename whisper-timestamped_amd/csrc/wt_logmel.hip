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
// radix-20 butterfly entirely in registers with the 20th roots of unity as
// scalar (SGPR) operands, the W400 twiddle comes from a 3.2 KB LDS table, the
// transposition between the two stages goes through LDS.  Three frames per
// wave, twelve per 256-thread workgroup, 250 workgroups per 30 s chunk.
// ~52 kFLOP per frame instead of 322 kFLOP for the direct DFT.
// Pass 1 writes log10(mel) and an atomic per-chunk max; pass 2 applies the
// clamp/scale and the zero padding.  Algorithmic bytes: 480000*4 read +
// n_mels*3000*4 written per chunk (the intermediate is re-read from L2).
#include <cmath>

#include "wt_common.h"

namespace wt {

__constant__ float k_c20[20];
__constant__ float k_s20[20];
__constant__ float k_hann[400];
__constant__ float2 k_w400[400];  // (cos, sin)(2*pi*k/400)

constexpr int FPB = 12;             // frames per workgroup
constexpr int SPAN = 160 * (FPB - 1) + 400;
constexpr int YP = 21;              // padded row of the stage-1 -> stage-2 exchange

__device__ __forceinline__ int enc_key(float v) {
    const int k = __float_as_int(v);
    return k >= 0 ? k : k ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float dec_key(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7FFFFFFF); }

constexpr int NNZ_CAP = 768;        // compact (banded) filterbank weights kept in LDS
constexpr int MAX_MELS = 256;

// Per-call preparation, ONE workgroup: reset the per-chunk max keys and turn the dense (n_mels x 201)
// filterbank into its banded form: lo[m], n[m], off[m] and the concatenated non-zero spans.
// ws layout (ints): keys[n_chunks] | lo[MAX_MELS] | n[MAX_MELS] | off[MAX_MELS] | total | weights[NNZ_CAP]
__global__ __launch_bounds__(256) void logmel_init_kernel(int *ws, const float *__restrict__ fb, int n_chunks, int n_mels) {
    __shared__ int s_lo[MAX_MELS], s_hi[MAX_MELS], s_off[MAX_MELS];
    const int tid = threadIdx.x;
    int *keys = ws, *lo = ws + n_chunks, *cnt = lo + MAX_MELS, *off = cnt + MAX_MELS, *total = off + MAX_MELS;
    float *wts = reinterpret_cast<float *>(total + 1);
    for (int t = tid; t < n_chunks; t += 256) keys[t] = (int)0x80000000;
    s_lo[tid] = 201;
    s_hi[tid] = 0;
    __syncthreads();
    for (int e = tid; e < n_mels * 201; e += 256) {
        if (fb[e] != 0.f) {
            const int m = e / 201, k = e - m * 201;
            atomicMin(&s_lo[m], k);
            atomicMax(&s_hi[m], k + 1);
        }
    }
    __syncthreads();
    const int width = (tid < n_mels && s_lo[tid] < s_hi[tid]) ? s_hi[tid] - s_lo[tid] : 0;
    if (width == 0) s_lo[tid] = 0;
    s_off[tid] = width;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan
        const int v = tid >= o ? s_off[tid - o] : 0;
        __syncthreads();
        s_off[tid] += v;
        __syncthreads();
    }
    const int my_off = s_off[tid] - width;
    if (tid < n_mels) { lo[tid] = s_lo[tid]; cnt[tid] = width; off[tid] = my_off; }
    const int tot = s_off[255];
    if (tid == 0) *total = tot;
    if (tot <= NNZ_CAP)
        for (int m = 0; m < n_mels; ++m) {  // block-cooperative copy of each band
            const int w = s_off[m] - (m ? s_off[m - 1] : 0), o = s_off[m] - w;
            for (int k = tid; k < w; k += 256) wts[o + k] = fb[m * 201 + s_lo[m] + k];
        }
}

__global__ __launch_bounds__(256) void stft_mel_kernel(const float *__restrict__ pcm, int64_t n_samples,
                                                       const int32_t *__restrict__ n_valid_samples,
                                                       const float *__restrict__ fb, const int *__restrict__ ws, int n_chunks,
                                                       int n_mels, int n_frames, float *__restrict__ mel_out) {
    __shared__ float span[SPAN];
    __shared__ float2 w400[400];
    __shared__ float2 yp[FPB][11][YP];   // stage-1 output, k1 = 0..10 (k1 > 10 is the conjugate of 20-k1)
    __shared__ float pw[FPB][204];
    __shared__ float fbw[NNZ_CAP];
    __shared__ int fb_lo[MAX_MELS], fb_n[MAX_MELS], fb_off[MAX_MELS];
    __shared__ int smax[4];

    const int chunk = blockIdx.y;
    const int f0 = blockIdx.x * FPB;
    const int nvs = n_valid_samples ? n_valid_samples[chunk] : (int)n_samples;
    const int nvf = min(nvs / 160, n_frames);  // frames kept after dropping the last stft frame
    if (f0 >= nvf) return;                    // whole tile is padding (block-uniform)
    const float *x = pcm + (int64_t)chunk * n_samples;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane / 20;       // frame slot inside the wave (0..2), lanes 60..63 idle
    const int u = lane - sub * 20;   // n2 in stage 1, k1 in stage 2
    const int slot = wave * 3 + sub;
    const bool act = lane < 60 && (f0 + slot) < nvf;

    // ---- phase A: every global read of the tile is issued here, one latency for all of them ----
    const int *g_lo = ws + n_chunks, *g_n = g_lo + MAX_MELS, *g_off = g_n + MAX_MELS, *g_tot = g_off + MAX_MELS;
    const float *g_w = reinterpret_cast<const float *>(g_tot + 1);
    const int nnz = *g_tot;
    const bool banded = nnz <= NNZ_CAP;
    float hw[20];                    // this lane's 20 window taps hann[20*n1 + u]
#pragma unroll
    for (int n1 = 0; n1 < 20; ++n1) hw[n1] = k_hann[20 * n1 + (lane < 60 ? u : 0)];
    for (int p = tid; p < SPAN; p += 256) {
        int i = f0 * 160 + p - 200;  // centre=True: frame f covers padded[160f, 160f+400)
        if (i < 0) i = -i;           // reflect (no edge repeat)
        if (i >= nvs) i = 2 * (nvs - 1) - i;
        i = max(0, min(i, nvs - 1));
        span[p] = x[i];
    }
    for (int p = tid; p < 400; p += 256) w400[p] = k_w400[p];
    if (tid < n_mels) { fb_lo[tid] = g_lo[tid]; fb_n[tid] = g_n[tid]; fb_off[tid] = g_off[tid]; }
    if (banded)
        for (int p = tid; p < nnz; p += 256) fbw[p] = g_w[p];
    __syncthreads();

    if (act) {
        // ---- stage 1: radix-20 over n1 for this lane's n2 = u ----
        float a[20];
        const float *fr = span + slot * 160;
#pragma unroll
        for (int n1 = 0; n1 < 20; ++n1) a[n1] = fr[20 * n1 + u] * hw[n1];
#pragma unroll
        for (int k1 = 0; k1 <= 10; ++k1) {
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) {
                sr = fmaf(a[n1], k_c20[(n1 * k1) % 20], sr);
                si = fmaf(a[n1], k_s20[(n1 * k1) % 20], si);
            }
            yp[slot][k1][u] = make_float2(sr, -si);
        }
    }
    __syncthreads();
    if (act) {
        // ---- stage 2: twiddle by W400^(n2*k1), radix-20 over n2 for this lane's k1 = u ----
        const int ks = u <= 10 ? u : 20 - u;
        const float cj = u <= 10 ? 1.f : -1.f;
        float br[20], bi[20];
#pragma unroll
        for (int n2 = 0; n2 < 20; ++n2) {
            const float2 v = yp[slot][ks][n2];
            const float2 w = w400[n2 * u];  // e^{-i t} = (cos t, -sin t)
            const float re = v.x, im = cj * v.y;
            br[n2] = re * w.x + im * w.y;
            bi[n2] = im * w.x - re * w.y;
        }
#pragma unroll
        for (int k2 = 0; k2 < 10; ++k2) {
            float xr = 0.f, xi = 0.f;
#pragma unroll
            for (int n2 = 0; n2 < 20; ++n2) {
                const float c = k_c20[(n2 * k2) % 20], s = k_s20[(n2 * k2) % 20];
                xr = fmaf(br[n2], c, fmaf(bi[n2], s, xr));
                xi = fmaf(bi[n2], c, fmaf(-br[n2], s, xi));
            }
            const float mag = sqrtf(xr * xr + xi * xi);  // torch: stft.abs() ** 2
            pw[slot][u + 20 * k2] = mag * mag;
        }
        if (u == 0) {  // k = 200: W20^(10 n2) = (-1)^n2
            float xr = 0.f, xi = 0.f;
#pragma unroll
            for (int n2 = 0; n2 < 20; ++n2) {
                xr += (n2 & 1) ? -br[n2] : br[n2];
                xi += (n2 & 1) ? -bi[n2] : bi[n2];
            }
            const float mag = sqrtf(xr * xr + xi * xi);
            pw[slot][200] = mag * mag;
        }
    }
    __syncthreads();

    // ---- mel projection + log10; thread -> (mel m, frame slot) ----
    float lmax = -INFINITY;
    for (int o = tid; o < n_mels * FPB; o += 256) {
        const int m = o / FPB, s = o - m * FPB;
        if (f0 + s >= nvf) continue;
        const int lo = fb_lo[m], n = fb_n[m];
        float acc = 0.f;
        if (banded) {
            const float *w = fbw + fb_off[m];
            for (int k = 0; k < n; ++k) acc = fmaf(w[k], pw[s][lo + k], acc);
        } else {
            const float *w = fb + m * 201 + lo;
            for (int k = 0; k < n; ++k) acc = fmaf(w[k], pw[s][lo + k], acc);
        }
        const float v = log10f(fmaxf(acc, 1e-10f));
        mel_out[((int64_t)chunk * n_mels + m) * n_frames + f0 + s] = v;
        lmax = fmaxf(lmax, v);
    }
    lmax = wave_max(lmax);
    if (lane == 0) smax[wave] = enc_key(lmax);
    __syncthreads();
    if (tid == 0) atomicMax(const_cast<int *>(ws) + chunk, max(max(smax[0], smax[1]), max(smax[2], smax[3])));
}

__global__ __launch_bounds__(256) void logmel_finalize_kernel(float *__restrict__ mel_out, const int *__restrict__ keys,
                                                              const int32_t *__restrict__ n_valid_samples, int64_t n_samples,
                                                              int n_mels, int n_frames, float *__restrict__ gmax) {
    const int chunk = blockIdx.y;
    const int nvs = n_valid_samples ? n_valid_samples[chunk] : (int)n_samples;
    const int nvf = min(nvs / 160, n_frames);
    const float mx = dec_key(keys[chunk]);
    const float floor_v = mx - 8.0f;
    float *base = mel_out + (int64_t)chunk * n_mels * n_frames;
    const int total = n_mels * n_frames;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int fr = e % n_frames;
        float v = 0.f;  // pad_or_trim: exact zeros
        if (fr < nvf) v = (fmaxf(base[e], floor_v) + 4.0f) / 4.0f;
        base[e] = v;
    }
    if (gmax && blockIdx.x == 0 && threadIdx.x == 0) gmax[chunk] = mx;
}

int scratch2(size_t bytes, void **out);

static int upload_tables(hipStream_t st) {
    static bool done = false;
    static float c20[20], s20[20], hann[400];
    static float2 w400[400];
    if (done) return WT_OK;
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < 20; ++k) {
        c20[k] = (float)std::cos(2.0 * PI * k / 20.0);
        s20[k] = (float)std::sin(2.0 * PI * k / 20.0);
    }
    for (int n = 0; n < 400; ++n) {
        hann[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * n / 400.0));
        w400[n] = make_float2((float)std::cos(2.0 * PI * n / 400.0), (float)std::sin(2.0 * PI * n / 400.0));
    }
    WT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(k_c20), c20, sizeof(c20), 0, hipMemcpyHostToDevice, st));
    WT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(k_s20), s20, sizeof(s20), 0, hipMemcpyHostToDevice, st));
    WT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(k_hann), hann, sizeof(hann), 0, hipMemcpyHostToDevice, st));
    WT_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(k_w400), w400, sizeof(w400), 0, hipMemcpyHostToDevice, st));
    WT_HIP(hipStreamSynchronize(st));  // once per process
    done = true;
    return WT_OK;
}

int logmel_batch(const float *pcm, int n_chunks, int64_t n_samples, const int32_t *n_valid_samples, const float *mel_fb,
                 int n_mels, int n_frames, float *mel_out, float *gmax, hipStream_t st) {
    if (!pcm || !mel_fb || !mel_out || n_chunks < 0 || n_samples < 201 || n_mels <= 0 || n_mels > MAX_MELS || n_frames <= 0) {
        set_error("wt_logmel_batch: bad argument");
        return WT_E_BADARG;
    }
    if (n_chunks == 0) return WT_OK;
    int rc = upload_tables(st);
    if (rc) return rc;
    int *ws = nullptr;
    rc = scratch2(((size_t)n_chunks + 3 * MAX_MELS + 1 + NNZ_CAP) * sizeof(int), (void **)&ws);
    if (rc) return rc;
    int *keys = ws;
    hipLaunchKernelGGL(logmel_init_kernel, dim3(1), dim3(256), 0, st, ws, mel_fb, n_chunks, n_mels);
    hipLaunchKernelGGL(stft_mel_kernel, dim3((n_frames + FPB - 1) / FPB, n_chunks), dim3(256), 0, st, pcm, n_samples,
                       n_valid_samples, mel_fb, ws, n_chunks, n_mels, n_frames, mel_out);
    const int total = n_mels * n_frames;
    int gx = (total + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(logmel_finalize_kernel, dim3(gx, n_chunks), dim3(256), 0, st, mel_out, keys, n_valid_samples, n_samples,
                       n_mels, n_frames, gmax);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
