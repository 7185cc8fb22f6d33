// NOT PART OF THE PRODUCT (not compiled by csrc/Makefile): the wave-autonomous, persistent variant of stft_mel_kernel
// that round 2 measured against the workgroup-tile kernel and did not keep.  It drops into wt_logmel.hip in front of
// logmel_finalize_kernel (it uses that file's tables and macros); the launch it needs is at the end of this file.
// Result on the MI355X (profiles/r2b_logmel_ab.jsonl, r2b_logmel_sq_counters.txt): same values bit for bit, 63 us
// against 60 us for 32 x 30 s chunks (97 vs 87 us with 128 mel bins).  Both kernels keep the VALU 62 % and the LDS
// array 55 % busy (34 % of the LDS cycles are bank conflicts); removing the four barriers per tile moved wave time from
// "parked at s_waitcnt / s_barrier" (53 % -> 42 %) to "issue stall" (21 % -> 31 %): the bound is the instruction
// streams themselves (745 VALU + ~110 LDS instructions per three frames), not the synchronisation.
// ---------------------------------------------------------------------------------------------------------------------
// Wave-autonomous version of the same arithmetic (stage 1 / stage 2 / mel projection are the statements above, value for
// value).  What changes is who waits for whom.  In stft_mel_kernel a workgroup walks one 12-frame tile through four
// barrier-separated phases, so every wave idles at four barriers per tile and the global reads of a tile are exposed
// once per tile (SQ counters of round 1: the SIMDs issue VALU 57 % of the time, SQ_WAIT_ANY 52 % of the wave cycles).
// Here a WAVE owns its three frames from PCM to log-mel:
//   * its own LDS slice (768 PCM samples + three stage-exchange slots, 8.6 KB); the three frames of a wave never touch
//     another wave's data, so the only synchronisation inside the loop is the in-order LDS queue of the wave itself
//     (compiler fences, no s_barrier);
//   * it is PERSISTENT over `ntw` consecutive triples of a chunk: the hann taps and the W400 twiddles of its lane live in
//     registers for the whole kernel (they depend on the lane only: 40 LDS reads per frame saved), the banded
//     filterbank is staged once per workgroup, and the PCM of triple t+1 is already in flight (3 x float4 per lane in
//     registers) while triple t is transformed;
//   * the power spectrum of a frame overwrites the frame's own stage-exchange slot (dead by then), and the mel projection
//     maps lane -> filter (lane and n_mels-1-lane: a narrow and a wide filter per lane balance the tap counts).
// 8 waves per workgroup, 2 workgroups per CU (72 KB of LDS each): 16 waves per CU as before, none of them at a barrier.
constexpr int WPB = 8;                       // waves per workgroup
constexpr int TSPAN = 768;                   // floats staged per triple (2 * 160 + 400 = 720 used; 3 float4 per lane)
constexpr int YSLOT = 11 * YP;               // float2 per stage-exchange slot (k1 = 0..10, n2 = 0..19 padded to 21)
struct __attribute__((aligned(16))) WaveLds {
    float span[TSPAN];
    float2 yp[3][YSLOT];
};

__global__ __launch_bounds__(64 * WPB, 4) void stft_mel_wave_kernel(
    const float *__restrict__ pcm, int64_t n_samples, const int32_t *__restrict__ n_valid_samples,
    const float *__restrict__ fb, const int *__restrict__ ws, float *__restrict__ wgmax, int n_mels, int n_frames,
    float *__restrict__ mel_out, int n_chunks, int tri_per_chunk, int groups_per_chunk, int ntw) {
    __shared__ WaveLds wl[WPB];
    __shared__ float2 w400[400];
    __shared__ float hann[400];
    __shared__ float fbw[NNZ_CAP];
    __shared__ unsigned char fb_lo[MAX_MELS], fb_n[MAX_MELS];
    __shared__ unsigned short fb_off[MAX_MELS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int *g_lo = ws, *g_n = g_lo + MAX_MELS, *g_off = g_n + MAX_MELS, *g_tot = g_off + MAX_MELS;
    const float *g_w = reinterpret_cast<const float *>(g_tot + 1);
    const int nnz = *g_tot;
    const bool banded = nnz <= NNZ_CAP;
    if (tid < n_mels) { fb_lo[tid] = (unsigned char)g_lo[tid]; fb_n[tid] = (unsigned char)g_n[tid]; fb_off[tid] = (unsigned short)g_off[tid]; }
    if (banded)
        for (int p = tid; p < nnz; p += 64 * WPB) fbw[p] = g_w[p];
    if (tid < 400) { w400[tid] = k_w400[tid]; hann[tid] = k_hann[tid]; }
    __syncthreads();                          // the only barrier of the kernel

    const int gw = blockIdx.x * WPB + wave;   // global wave index -> (chunk, group of ntw triples)
    const int chunk = gw / groups_per_chunk;
    const int group = gw - chunk * groups_per_chunk;
    if (chunk >= n_chunks) return;            // wave-uniform
    const int nvs = n_valid_samples ? n_valid_samples[chunk] : (int)n_samples;
    const int nvf = min(nvs / 160, n_frames);  // frames kept after dropping the last stft frame
    const int t_begin = group * ntw;
    const int t_end = min(min(t_begin + ntw, tri_per_chunk), (nvf + 2) / 3);   // triples that hold at least one real frame
    float *wg_slot = wgmax + (size_t)chunk * groups_per_chunk + group;
    if (t_begin >= t_end) {
        if (lane == 0) *wg_slot = -INFINITY;
        return;
    }
    const float *x = pcm + (int64_t)chunk * n_samples;
    const bool vec_ok = (n_samples & 3) == 0 && (reinterpret_cast<uintptr_t>(pcm) & 15) == 0;
    const int sub = lane / 20;                // frame slot inside the wave (0..2), lanes 60..63 idle
    const int u = lane - sub * 20;            // n2 in stage 1, k1 in stage 2
    float *span = wl[wave].span;
    float2 *ypw = wl[wave].yp[sub < 3 ? sub : 0];

    // per-lane constants, fetched once: the hann taps 20*n1 + u of this lane (the W400 twiddles stay in an LDS table:
    // keeping those 40 values in registers as well spills at 4 waves per SIMD)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const float *hn = hann + u;

    // PCM of one triple: samples [160*3t - 200, +720) of the centre-padded signal.  Interior triples are three
    // 16-byte loads per lane; the first / last ones of a chunk pay for the reflect arithmetic.
    // (global_load_lds_dwordx4: HBM/L2 -> LDS without a VGPR round trip, lane l of a load lands at base + 16*l;
    // completion is tracked by vmcnt)
    auto interior = [&](int t) { const int i0 = 480 * t - 200; return vec_ok && i0 >= 0 && i0 + TSPAN <= nvs; };
    auto fetch = [&](int t) {
        const float *src = x + (480 * t - 200) + 4 * lane;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 256 * k),
                                             (__attribute__((address_space(3))) void *)(span + 256 * k), 16, 0, 0);
    };
    auto gather = [&](int t) {
        const int i0 = 480 * t - 200;
        for (int p = lane; p < 720; p += 64) {
            int i = i0 + p;
            if (i < 0) i = -i;                 // reflect (no edge repeat)
            if (i >= nvs) i = 2 * (nvs - 1) - i;
            i = max(0, min(i, nvs - 1));
            span[p] = x[i];
        }
    };

    if (interior(t_begin)) { fetch(t_begin); wait_vmcnt0(); } else gather(t_begin);
    wave_lds_fence();

    float lmax = -INFINITY;
    for (int t = t_begin; t < t_end; ++t) {
        const int f0 = 3 * t;
        const bool more = t + 1 < t_end;
        const bool nx_ok = more && interior(t + 1);   // wave-uniform
        const bool act = lane < 60 && (f0 + sub) < nvf;

        if (act) {
            // ---- stage 1 (as in stft_mel_kernel) ----
            float a[20];
            const float *fr = span + sub * 160;
#pragma unroll
            for (int n1 = 0; n1 < 20; ++n1) a[n1] = fr[20 * n1 + u] * hn[20 * n1];
            float ep[10], em[10];
#pragma unroll
            for (int n = 1; n < 10; ++n) {
                ep[n] = a[n] + a[20 - n];
                em[n] = a[n] - a[20 - n];
            }
            const float base_e = a[0] + a[10], base_o = a[0] - a[10];
            float sr[11], si[11];
#define WT_S1(K)                                                                                          \
            {                                                                                             \
                float Ae = 0.f, Ao = 0.f, Be = 0.f, Bo = 0.f;                                             \
                WT_MAC(Ae, ep[2], c20(2 * K)); WT_MAC(Ae, ep[4], c20(4 * K)); WT_MAC(Ae, ep[6], c20(6 * K));  \
                WT_MAC(Ae, ep[8], c20(8 * K));                                                            \
                WT_MAC(Ao, ep[1], c20(1 * K)); WT_MAC(Ao, ep[3], c20(3 * K)); WT_MAC(Ao, ep[5], c20(5 * K));  \
                WT_MAC(Ao, ep[7], c20(7 * K)); WT_MAC(Ao, ep[9], c20(9 * K));                             \
                WT_MAC(Be, em[2], s20(2 * K)); WT_MAC(Be, em[4], s20(4 * K)); WT_MAC(Be, em[6], s20(6 * K));  \
                WT_MAC(Be, em[8], s20(8 * K));                                                            \
                WT_MAC(Bo, em[1], s20(1 * K)); WT_MAC(Bo, em[3], s20(3 * K)); WT_MAC(Bo, em[5], s20(5 * K));  \
                WT_MAC(Bo, em[7], s20(7 * K)); WT_MAC(Bo, em[9], s20(9 * K));                             \
                const float base = (K & 1) ? base_o : base_e;                                             \
                sr[K] = base + Ae + Ao; sr[10 - K] = base + Ae - Ao;                                      \
                si[K] = Be + Bo;        si[10 - K] = Bo - Be;                                             \
            }
            WT_S1(0) WT_S1(1) WT_S1(2) WT_S1(3) WT_S1(4) WT_S1(5)
#undef WT_S1
#pragma unroll
            for (int k1 = 0; k1 <= 10; ++k1) ypw[k1 * YP + u] = make_float2(sr[k1], -si[k1]);
        }
        wave_lds_fence();                              // the frame's 20 lanes exchange through the wave's own LDS slice
        // The PCM of this triple has been consumed (its reads fed the arithmetic above): the next triple's samples
        // stream into the same buffer while stage 2 and the mel projection run.
        if (nx_ok) fetch(t + 1);
        float pk2[11];
        if (act) {
            // ---- stage 2 (as in stft_mel_kernel; twiddles from registers) ----
            const int ks = u <= 10 ? u : 20 - u;
            const f2 cjv = u <= 10 ? (f2){1.f, 1.f} : (f2){1.f, -1.f};
            f2 B[20];
            int woff = 0;                        // byte offset of W400^(n2*u) in the table
            const int wstep = u * (int)sizeof(float2);
#pragma unroll
            for (int n2 = 0; n2 < 20; ++n2) {
                const float2 vv = ypw[ks * YP + n2];
                const float2 ww = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(w400) + woff);
                woff += wstep;
                asm volatile("" : "+v"(woff));   // keep it an add (unrolled, hipcc turns it into a v_mul_lo_u32)
                const f2 v = (f2){vv.x, vv.y} * cjv;
                const f2 w = (f2){ww.x, ww.y};
                f2 b = v * (f2){w.x, w.x};
                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(b) : "v"(v), "v"(w));
                B[n2] = b;
            }
            f2 P[10], D[10];
#pragma unroll
            for (int n = 1; n < 10; ++n) {
                P[n] = B[n] + B[20 - n];
                D[n] = B[n] - B[20 - n];
            }
            const f2 Bev = B[0] + B[10], Bod = B[0] - B[10];
#define WT_MAC2(acc, x, w)                                                          \
    do {                                                                            \
        constexpr float _w = (w);                                                   \
        if (_w == 1.0f) acc += (x);                                                 \
        else if (_w == -1.0f) acc -= (x);                                           \
        else if (_w != 0.0f) acc = __builtin_elementwise_fma((x), (f2){_w, _w}, acc); \
    } while (0)
#define WT_S2(K)                                                                                              \
            {                                                                                                 \
                f2 CDe = {0.f, 0.f}, CDo = {0.f, 0.f}, SDe = {0.f, 0.f}, SDo = {0.f, 0.f};                    \
                WT_MAC2(CDe, P[2], c20(2 * K)); WT_MAC2(CDe, P[4], c20(4 * K)); WT_MAC2(CDe, P[6], c20(6 * K));   \
                WT_MAC2(CDe, P[8], c20(8 * K));                                                               \
                WT_MAC2(CDo, P[1], c20(1 * K)); WT_MAC2(CDo, P[3], c20(3 * K)); WT_MAC2(CDo, P[5], c20(5 * K));   \
                WT_MAC2(CDo, P[7], c20(7 * K)); WT_MAC2(CDo, P[9], c20(9 * K));                               \
                WT_MAC2(SDe, D[2], s20(2 * K)); WT_MAC2(SDe, D[4], s20(4 * K)); WT_MAC2(SDe, D[6], s20(6 * K));   \
                WT_MAC2(SDe, D[8], s20(8 * K));                                                               \
                WT_MAC2(SDo, D[1], s20(1 * K)); WT_MAC2(SDo, D[3], s20(3 * K)); WT_MAC2(SDo, D[5], s20(5 * K));   \
                WT_MAC2(SDo, D[7], s20(7 * K)); WT_MAC2(SDo, D[9], s20(9 * K));                               \
                const f2 Bk = (K & 1) ? Bod : Bev;                                                            \
                const f2 U0 = Bk + (CDe + CDo), V0 = SDe + SDo, U1 = Bk + (CDe - CDo), V1 = SDo - SDe;        \
                const float r0 = U0.x + V0.y, i0 = U0.y - V0.x, r1 = U1.x + V1.y, i1 = U1.y - V1.x;          \
                pk2[K] = r0 * r0 + i0 * i0; pk2[10 - K] = r1 * r1 + i1 * i1;                                  \
            }
            WT_S2(0) WT_S2(1) WT_S2(2) WT_S2(3) WT_S2(4) WT_S2(5)
#undef WT_S2
#undef WT_MAC2
        }
        // the frame's power spectrum goes where its stage-1 output was: every lane of the wave has issued its reads of
        // that slot (above, in program order; the LDS queue of a wave is served in order)
        wave_lds_fence();
        if (act) {
            float *pw = reinterpret_cast<float *>(ypw);
#pragma unroll
            for (int k2 = 0; k2 < 10; ++k2) pw[u + 20 * k2] = pk2[k2];
            if (u == 0) pw[200] = pk2[10];   // k = 200 (k1 = 0, k2 = 10)
        }
        // The next triple's PCM has had all of stage 2 to arrive.  Waiting for it HERE, before the stores of the mel
        // projection are issued, keeps those stores out of the wait (vmcnt counts stores too, and cannot be waited on
        // selectively once loads and stores are mixed).
        if (nx_ok) wait_vmcnt0();
        wave_lds_fence();

        // ---- mel projection + log10: lane -> filter lo + lane and filter hi - lane (a narrow and a wide one), the
        //      three frames of the triple share every tap read ----
        const float *p0 = reinterpret_cast<const float *>(wl[wave].yp[0]);
        const float *p1 = reinterpret_cast<const float *>(wl[wave].yp[1]);
        const float *p2 = reinterpret_cast<const float *>(wl[wave].yp[2]);
        const int left = nvf - f0;                 // >= 1: frames of this triple that exist
        for (int lo_m = 0, hi_m = n_mels - 1; lo_m <= hi_m; lo_m += 64, hi_m -= 64) {
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int m = side == 0 ? lo_m + lane : hi_m - lane;
                const bool ok = side == 0 ? m <= hi_m : m >= lo_m + 64;
                if (!ok) continue;
                const int lo = fb_lo[m], n = fb_n[m];
                const float *w = banded ? fbw + fb_off[m] : fb + m * 201 + lo;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                for (int k = 0; k < n; ++k) {
                    const float wk = w[k];
                    a0 = fmaf(wk, p0[lo + k], a0);
                    a1 = fmaf(wk, p1[lo + k], a1);
                    a2 = fmaf(wk, p2[lo + k], a2);
                }
                const float L2 = 0.30102999566398120f;
                const float v0 = __builtin_amdgcn_logf(fmaxf(a0, 1e-10f)) * L2, v1 = __builtin_amdgcn_logf(fmaxf(a1, 1e-10f)) * L2;
                const float v2 = __builtin_amdgcn_logf(fmaxf(a2, 1e-10f)) * L2;
                float *out = mel_out + ((int64_t)chunk * n_mels + m) * n_frames + f0;
                out[0] = v0;
                lmax = fmaxf(lmax, v0);
                if (left > 1) { out[1] = v1; lmax = fmaxf(lmax, v1); }
                if (left > 2) { out[2] = v2; lmax = fmaxf(lmax, v2); }
            }
        }
        wave_lds_fence();                              // the slots are free again
        if (more && !nx_ok) {
            gather(t + 1);
            wave_lds_fence();
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) *wg_slot = lmax;
}


// launch (host side):
//   const int tri_per_chunk = (n_frames + 2) / 3;
//   int ntw = clamp(ceil(n_chunks * tri_per_chunk / 4096), 1, 8);             // about one resident set of waves
//   const int groups_per_chunk = ceil(tri_per_chunk / ntw);                   // = n_wg of logmel_finalize_kernel
//   stft_mel_wave_kernel<<<ceil(n_chunks * groups_per_chunk / WPB), 64 * WPB, 0, stream>>>(pcm, n_samples,
//       n_valid_samples, mel_fb, ws, wgmax, n_mels, n_frames, mel_out, n_chunks, tri_per_chunk, groups_per_chunk, ntw);
