// Confidence path: chosen-token log-probability of each decode step, and the
// padding detector.
//
// logprob_gather replaces /root/reference/whisper_timestamped/transcribe.py
//   :871-876  logits -> (logit filters write -inf) -> F.log_softmax(dim=-1)
//   :735      logprob[tok] gather                      (efficient strategy)
//   :1245     F.log_softmax(logits, dim=-1); :1292 logprobs[:, step, tok]  (naive)
// without materialising the (n_rows, V) log-prob matrix: one workgroup per
// row streams the V logits ONCE from HBM (16-byte loads on the aligned body),
// keeps a running (max, sum-of-exp) pair per thread, merges the pairs across
// the workgroup and writes a single float.  Algorithmic bytes: V*4 per row.
//
// find_start_padding replaces transcribe.py:1795-1805.
#include <hip/hip_fp16.h>

#include "wt_common.h"

namespace wt {

// Running (max, sum of exp(x - max)) of one thread.  `m` starts at a FINITE sentinel, not at -inf: every update below is
// then branch-free -- x - m is -inf for a suppressed logit (-inf) and never inf - inf -- and what is left per element is
// a packed subtract, a packed multiply by log2(e), v_exp_f32 and a packed add (round 6: 3.4 VALU instructions per logit
// where the branching form with a compensated product and a rescale test every four logits took 14.4 -- the stream
// is HBM-bound either way, but the VALU slots it no longer takes are the ones the kernels that run beside it under
// the hilo schedule, stft_mel and dtw_kernel, are bound by).  The product (x - m) * log2(e) is rounded once: half an ulp
// of the EXPONENT, i.e. a relative error of the term that grows with |x - m| exactly as the term's weight in the sum
// vanishes (measured: |d log-sum-exp| < 1e-6 for N(0, sigma) rows, sigma = 1 .. 12, V = 51865).
struct MS {
    float m, s;
};
constexpr float MS_NONE = -3.0e38f;                                  // "nothing seen yet" (finite)
constexpr float L2E = 1.44269502162933349609375f;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }   // v_max3_f32
// the new maximum enters: s *= exp(m_old - m_new) (exp2(-inf) = 0 retires the sentinel; exact 1 when nothing changes)
__device__ __forceinline__ void ms_raise(MS &a, float cm) {
    const float mn = fmaxf(a.m, cm);
    a.s *= __builtin_amdgcn_exp2f((a.m - mn) * L2E);
    a.m = mn;
}
__device__ __forceinline__ f2 exp_pair(f2 x, f2 m2) {
    const f2 y = (x - m2) * (f2){L2E, L2E};
    return (f2){__builtin_amdgcn_exp2f(y.x), __builtin_amdgcn_exp2f(y.y)};
}
__device__ __forceinline__ void ms_add4(MS &a, f4 r) {
    ms_raise(a, fmaxf(max3(r.x, r.y, r.z), r.w));
    const f2 m2 = (f2){a.m, a.m};
    const f2 e = exp_pair(r.xy, m2) + exp_pair(r.zw, m2);
    a.s += e.x + e.y;
}
// sixteen logits (the four 16-byte loads a thread keeps in flight): ONE maximum update for all of them
__device__ __forceinline__ void ms_add16(MS &a, f4 r0, f4 r1, f4 r2, f4 r3) {
    const float c0 = max3(r0.x, r0.y, r0.z), c1 = max3(r0.w, r1.x, r1.y), c2 = max3(r1.z, r1.w, r2.x);
    const float c3 = max3(r2.y, r2.z, r2.w), c4 = max3(r3.x, r3.y, r3.z);
    ms_raise(a, fmaxf(max3(c0, c1, c2), max3(c3, c4, r3.w)));
    const f2 m2 = (f2){a.m, a.m};
    const f2 e0 = exp_pair(r0.xy, m2) + exp_pair(r0.zw, m2), e1 = exp_pair(r1.xy, m2) + exp_pair(r1.zw, m2);
    const f2 e2 = exp_pair(r2.xy, m2) + exp_pair(r2.zw, m2), e3 = exp_pair(r3.xy, m2) + exp_pair(r3.zw, m2);
    const f2 e = (e0 + e1) + (e2 + e3);
    a.s += e.x + e.y;
}
__device__ __forceinline__ void ms_add1(MS &a, float x) {
    ms_raise(a, x);
    a.s += __builtin_amdgcn_exp2f((x - a.m) * L2E);
}
// after the stream: the sentinel becomes the -inf the merges below test for
__device__ __forceinline__ MS ms_close(MS a) {
    if (a.m == MS_NONE) a.m = -INFINITY;
    return a;
}
__device__ __forceinline__ MS ms_merge(MS a, MS b) {
    if (b.m == -INFINITY) return a;
    if (a.m == -INFINITY) return b;
    MS r;
    r.m = fmaxf(a.m, b.m);
    r.s = a.s * expf(a.m - r.m) + b.s * expf(b.m - r.m);
    return r;
}

__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const __half *p) { return __half2float(*p); }

template <typename LT>
__global__ __launch_bounds__(256) void logprob_gather_kernel(const LT *__restrict__ logits, int64_t row_stride, int V,
                                                             const int32_t *__restrict__ token,
                                                             const uint8_t *__restrict__ suppress, int suppress_rows,
                                                             const int32_t *__restrict__ row_index,
                                                             float *__restrict__ out) {
    const int row = row_index ? row_index[blockIdx.x] : blockIdx.x;   // which logits row this output reads
    const LT *x = logits + (int64_t)row * row_stride;
    const uint8_t *sup = suppress ? suppress + (suppress_rows > 1 ? (int64_t)row * V : 0) : nullptr;
    const int tid = threadIdx.x;
    MS acc = {MS_NONE, 0.f};

    constexpr int VEC = 16 / sizeof(LT);  // elements per 16-byte load
    // The vector stream starts on a 128-BYTE boundary (not merely a 16-byte one): a wave's 64 x 16-byte request is
    // then exactly eight cache lines.  Started on an arbitrary 16-byte boundary it straddles nine, and the line it
    // shares with the neighbouring wave is fetched twice -- the stream is non-temporal, nothing keeps it in L2 --
    // which was 6.5 % more HBM bytes than the rows hold (FETCH_SIZE: 1583 MB for 1487 MB of logits).
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    int head = (int)(((128 - (addr & 127)) & 127) / sizeof(LT));   // <= 31 (fp32) / 63 (fp16) elements: one per thread
    if (head > V) head = V;
    const int nvec = (V - head) / VEC;
    const int tail0 = head + nvec * VEC;

    if (tid < head) ms_add1(acc, (sup && sup[tid]) ? -INFINITY : ldf(x + tid));
    if (tid < V - tail0) {
        const int e = tail0 + tid;
        ms_add1(acc, (sup && sup[e]) ? -INFINITY : ldf(x + e));
    }
    if constexpr (sizeof(LT) == 4) {
        const f4 *xv = reinterpret_cast<const f4 *>(x + head);
        int v = tid;
        if (!sup) {
            // The common, unmasked stream.  Every logit is read exactly once: non-temporal 16-byte loads, four in
            // flight per thread (read-only probe on this part: 6.6 TB/s this way against 6.0 with plain loads,
            // tools/probes/read_probe.hip).
            const f4 *xn = reinterpret_cast<const f4 *>(x + head);
            for (; v + 768 < nvec; v += 1024) {
                const f4 r0 = __builtin_nontemporal_load(xn + v), r1 = __builtin_nontemporal_load(xn + v + 256);
                const f4 r2 = __builtin_nontemporal_load(xn + v + 512), r3 = __builtin_nontemporal_load(xn + v + 768);
                ms_add16(acc, r0, r1, r2, r3);
            }
        }
        for (; v < nvec; v += 256) {
            f4 r = xv[v];
            if (sup) {
                const uint8_t *sp = sup + head + 4 * v;
                if (sp[0]) r.x = -INFINITY;
                if (sp[1]) r.y = -INFINITY;
                if (sp[2]) r.z = -INFINITY;
                if (sp[3]) r.w = -INFINITY;
            }
            ms_add4(acc, r);
        }
    } else {
        const uint4 *xv = reinterpret_cast<const uint4 *>(x + head);
        for (int v = tid; v < nvec; v += 256) {
            const uint4 raw = xv[v];
            const __half2 *h = reinterpret_cast<const __half2 *>(&raw);
            float f[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float2 t = __half22float2(h[k]);
                f[2 * k] = t.x;
                f[2 * k + 1] = t.y;
            }
            if (sup) {
                const uint8_t *sp = sup + head + 8 * v;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (sp[k]) f[k] = -INFINITY;
            }
            ms_add4(acc, (f4){f[0], f[1], f[2], f[3]});
            ms_add4(acc, (f4){f[4], f[5], f[6], f[7]});
        }
    }
    // wave butterfly, then across the 4 waves through LDS
    acc = ms_close(acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MS b;
        b.m = __shfl_xor(acc.m, o, 64);
        b.s = __shfl_xor(acc.s, o, 64);
        acc = ms_merge(acc, b);
    }
    __shared__ MS part[4];
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        MS t = ms_merge(ms_merge(part[0], part[1]), ms_merge(part[2], part[3]));
        const int tok = token[blockIdx.x];
        float xt = -INFINITY;
        if (tok >= 0 && tok < V && !(sup && sup[tok])) xt = ldf(x + tok);
        out[blockIdx.x] = (xt - t.m) - logf(t.s);
    }
}

// row_index (optional): output r reads logits row row_index[r] (rows may repeat or be skipped: the batched naive
// strategy gathers the text positions of many padded windows from one (B*T_max, V) block); per-row suppress masks
// are then indexed by the logits row as well.
int logprob_gather_batch(const void *logits, int dtype, int64_t row_stride, int n_rows, int V, const int32_t *token,
                         const uint8_t *suppress, int suppress_rows, const int32_t *row_index, float *out, hipStream_t st) {
    if (!logits || !token || !out || n_rows < 0 || V <= 0 || row_stride < V ||
        (suppress && suppress_rows != 1 && suppress_rows != n_rows && !row_index)) {
        set_error("wt_logprob_gather_batch: bad argument");
        return WT_E_BADARG;
    }
    if (n_rows == 0) return WT_OK;
    const uint8_t *sup = suppress_rows > 0 ? suppress : nullptr;
    if (dtype == WT_DTYPE_F32)
        hipLaunchKernelGGL(logprob_gather_kernel<float>, dim3(n_rows), dim3(256), 0, st, (const float *)logits, row_stride, V,
                           token, sup, suppress_rows, row_index, out);
    else if (dtype == WT_DTYPE_F16)
        hipLaunchKernelGGL(logprob_gather_kernel<__half>, dim3(n_rows), dim3(256), 0, st, (const __half *)logits, row_stride,
                           V, token, sup, suppress_rows, row_index, out);
    else {
        set_error("wt_logprob_gather_batch: dtype=%d", dtype);
        return WT_E_BADARG;
    }
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// ---------------------------------------------------------------------------
// B decoder streams (whisper_timestamped/streams.py): what the replayed hook state machine can still ask of a decoder
// call's filtered logits row, taken when the row is final (the sampler has filtered it in place and the sampled token
// is known) -- instead of keeping the row itself in a (streams, 449, V) ring (93 MB per stream):
//   transcribe.py:735,876  log_softmax(row)[sampled token]                     -> rec[0]
//   transcribe.py:508,729,879  argmax(row)                                      -> rec[3]  (first index of the maximum)
//   log_softmax(row)[t] for the other tokens the reference's fallbacks name     -> (x[t] - rec[1]) - rec[2] with
//        x[t] from rec[4..7] (eot, <|notimestamps|> ...) or from the kept slice (any timestamp token)
//   transcribe.py:535  argmax(row[start_token + 1:]), start_token a timestamp   -> the kept slice row[slice_begin:]
// One workgroup per row, one streaming pass (the same 128-byte-aligned non-temporal stream as logprob_gather_kernel,
// same accumulation order: rec[0] is bit for bit what wt_logprob_gather_batch returns for that row and token).
struct Best {
    float v;
    int i;
};
__device__ __forceinline__ void best_add(Best &b, float x, int i) {
    if (x > b.v || (x == b.v && i < b.i)) {
        b.v = x;
        b.i = i;
    }
}

template <typename TT>
__global__ __launch_bounds__(256) void logprob_digest_kernel(const float *__restrict__ logits, int64_t row_stride, int V,
                                                             const TT *__restrict__ token, int64_t token_stride,
                                                             const int32_t *__restrict__ ring_index, int64_t ring_rows,
                                                             int64_t ring_row, int4 aux, int n_aux, int slice_begin,
                                                             float *__restrict__ digest, float *__restrict__ slice) {
    const float *x = logits + (int64_t)blockIdx.x * row_stride;
    const int tid = threadIdx.x;
    MS acc = {MS_NONE, 0.f};
    Best best = {-INFINITY, 0x7fffffff};
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    int head = (int)(((128 - (addr & 127)) & 127) / sizeof(float));
    if (head > V) head = V;
    const int nvec = (V - head) / 4;
    const int tail0 = head + nvec * 4;
    if (tid < head) {
        const float v = x[tid];
        ms_add1(acc, v);
        best_add(best, v, tid);
    }
    if (tid < V - tail0) {
        const float v = x[tail0 + tid];
        ms_add1(acc, v);
        best_add(best, v, tail0 + tid);
    }
    const f4 *xn = reinterpret_cast<const f4 *>(x + head);
    int v = tid;
#define WT_BEST4(r, vi)                                     \
    {                                                       \
        const int e = head + 4 * (vi);                      \
        best_add(best, r.x, e);                             \
        best_add(best, r.y, e + 1);                         \
        best_add(best, r.z, e + 2);                         \
        best_add(best, r.w, e + 3);                         \
    }
    for (; v + 768 < nvec; v += 1024) {
        const f4 r0 = __builtin_nontemporal_load(xn + v), r1 = __builtin_nontemporal_load(xn + v + 256);
        const f4 r2 = __builtin_nontemporal_load(xn + v + 512), r3 = __builtin_nontemporal_load(xn + v + 768);
        ms_add16(acc, r0, r1, r2, r3);
        WT_BEST4(r0, v) WT_BEST4(r1, v + 256) WT_BEST4(r2, v + 512) WT_BEST4(r3, v + 768)
    }
    for (; v < nvec; v += 256) {
        const f4 r = xn[v];
        ms_add4(acc, r);
        WT_BEST4(r, v)
    }
#undef WT_BEST4
    acc = ms_close(acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        MS b;
        b.m = __shfl_xor(acc.m, o, 64);
        b.s = __shfl_xor(acc.s, o, 64);
        acc = ms_merge(acc, b);
        const float bv = __shfl_xor(best.v, o, 64);
        const int bi = __shfl_xor(best.i, o, 64);
        best_add(best, bv, bi);
    }
    __shared__ MS part[4];
    __shared__ Best bpart[4];
    if ((tid & 63) == 0) {
        part[tid >> 6] = acc;
        bpart[tid >> 6] = best;
    }
    __syncthreads();
    const int64_t rec_index = (int64_t)ring_index[blockIdx.x] * ring_rows + ring_row;
    if (tid == 0) {
        MS t = ms_merge(ms_merge(part[0], part[1]), ms_merge(part[2], part[3]));
        Best b = bpart[0];
        best_add(b, bpart[1].v, bpart[1].i);
        best_add(b, bpart[2].v, bpart[2].i);
        best_add(b, bpart[3].v, bpart[3].i);
        if (b.i == 0x7fffffff) b.i = 0;      // a row of -inf / NaN only: torch.argmax's answer for it
        const int64_t tok = (int64_t)token[(int64_t)blockIdx.x * token_stride];
        float xt = -INFINITY;
        if (tok >= 0 && tok < V) xt = x[tok];
        float *rec = digest + rec_index * 8;
        const float ls = logf(t.s);
        rec[0] = (xt - t.m) - ls;
        rec[1] = t.m;
        rec[2] = ls;
        rec[3] = __int_as_float(b.i);
        const int a[4] = {aux.x, aux.y, aux.z, aux.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[4 + k] = (k < n_aux && a[k] >= 0 && a[k] < V) ? x[a[k]] : -INFINITY;
    }
    if (slice) {
        const int n = V - slice_begin;
        float *dst = slice + rec_index * n;
        for (int e = tid; e < n; e += 256) dst[e] = x[slice_begin + e];
    }
}

int logprob_digest_streams(const float *logits, int64_t row_stride, int n_rows, int V, const void *token, int token_dtype,
                           int64_t token_stride, const int32_t *ring_index, int64_t ring_rows, int64_t ring_row,
                           const int32_t *aux_host, int n_aux, int slice_begin, float *digest, float *slice, hipStream_t st) {
    if (!logits || !token || !ring_index || !digest || n_rows < 0 || V <= 0 || row_stride < V || ring_rows <= 0 || ring_row < 0 ||
        ring_row >= ring_rows || n_aux < 0 || n_aux > 4 || (n_aux > 0 && !aux_host) || (slice && (slice_begin < 0 || slice_begin >= V)) ||
        (token_dtype != 0 && token_dtype != 1)) {
        set_error("wt_logprob_digest_streams: bad argument");
        return WT_E_BADARG;
    }
    if (n_rows == 0) return WT_OK;
    int4 aux = make_int4(-1, -1, -1, -1);
    if (n_aux > 0) aux.x = aux_host[0];
    if (n_aux > 1) aux.y = aux_host[1];
    if (n_aux > 2) aux.z = aux_host[2];
    if (n_aux > 3) aux.w = aux_host[3];
    if (token_dtype == 0)
        hipLaunchKernelGGL(logprob_digest_kernel<int32_t>, dim3(n_rows), dim3(256), 0, st, logits, row_stride, V,
                           (const int32_t *)token, token_stride, ring_index, ring_rows, ring_row, aux, n_aux, slice_begin, digest, slice);
    else
        hipLaunchKernelGGL(logprob_digest_kernel<int64_t>, dim3(n_rows), dim3(256), 0, st, logits, row_stride, V,
                           (const int64_t *)token, token_stride, ring_index, ring_rows, ring_row, aux, n_aux, slice_begin, digest, slice);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

// ---------------------------------------------------------------------------
// transcribe.py:1795-1805.  One workgroup per (n_mels, n_cols) window.  Like
// the reference it looks at the last column first (80 floats: the common "no
// padding" answer costs one tiny read) and only then walks BACKWARDS, 256
// columns per trip with every mel row read coalesced, stopping at the first
// tile that holds a non-zero column -- the bytes touched are proportional to
// the length of the padding, not to the window.
__global__ __launch_bounds__(256) void find_start_padding_kernel(const float *__restrict__ mel, int n_mels, int n_cols,
                                                                 int32_t *__restrict__ out) {
    const float *m = mel + (int64_t)blockIdx.x * n_mels * n_cols;
    const int tid = threadIdx.x;
    __shared__ int s_red[4];
    __shared__ int s_flag;
    // 1. last column all exactly zero?  (min == max == 0 in the reference; -0.0 == 0, NaN != 0)
    int nz = 0;
    for (int r = tid; r < n_mels; r += 256) nz |= !(m[(int64_t)r * n_cols + n_cols - 1] == 0.f);
    nz = wave_max_i(nz);
    if ((tid & 63) == 0) s_red[tid >> 6] = nz;
    __syncthreads();
    if (tid == 0) s_flag = s_red[0] | s_red[1] | s_red[2] | s_red[3];
    __syncthreads();
    if (s_flag) {
        if (tid == 0) out[blockIdx.x] = -1;
        return;
    }
    // 2. highest column in [1, n_cols-2] that differs from zero
    for (int hi = n_cols - 2; hi >= 1; hi -= 256) {
        const int c = hi - tid;
        int found = 0;
        if (c >= 1) {
            bool any = false;
#pragma unroll 8
            for (int r = 0; r < n_mels; ++r) any |= !(m[(int64_t)r * n_cols + c] == 0.f);
            found = any ? c : 0;
        }
        found = wave_max_i(found);
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = found;
        __syncthreads();
        const int best = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
        if (best > 0) {
            if (tid == 0) out[blockIdx.x] = best + 1;
            return;
        }
    }
    if (tid == 0) out[blockIdx.x] = 0;
}

int find_start_padding_batch(const float *mel, int n_chunks, int n_mels, int n_cols, int32_t *out, hipStream_t st) {
    if (!mel || !out || n_chunks < 0 || n_mels <= 0 || n_cols < 2) {
        set_error("wt_find_start_padding_batch: bad argument");
        return WT_E_BADARG;
    }
    if (n_chunks == 0) return WT_OK;
    hipLaunchKernelGGL(find_start_padding_kernel, dim3(n_chunks), dim3(256), 0, st, mel, n_mels, n_cols, out);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
