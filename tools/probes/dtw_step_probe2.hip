// Micro-probe 2: candidate implementations of the off-chain parts of a DTW sweep step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 8192
__device__ __forceinline__ void shift_in(double &up, double g) {
    union { double d; int i[2]; } s, o; s.d = g; o.d = up;
    o.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xf, 0xf, false);
    o.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xf, 0xf, false);
    up = o.d;
}
__device__ __forceinline__ void plane_bit(uint32_t &w, double a, double b) {  // w = 2w + (a < b)
    asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void push_lane63(double &acc, double best) {  // acc lane0 <- best[63]; acc[l] <- acc[l-1]
    union { double d; int i[2]; } b, t, a; b.d = best; a.d = acc;
    t.i[0] = __builtin_amdgcn_mov_dpp(b.i[0], 0x13C, 0xf, 0xf, false);  // wave_ror:1
    t.i[1] = __builtin_amdgcn_mov_dpp(b.i[1], 0x13C, 0xf, 0xf, false);
    t.i[0] = __builtin_amdgcn_update_dpp(t.i[0], a.i[0], 0x138, 0xf, 0xf, false);
    t.i[1] = __builtin_amdgcn_update_dpp(t.i[1], a.i[1], 0x138, 0xf, 0xf, false);
    acc = t.d;
}
template <int MODE>
__global__ void probe(double *out, long long *cyc, const float *cst, uint32_t *planes) {
    __shared__ double lds[4096];
    double g = threadIdx.x * 1e-3, u0 = 1e300, u1 = 1e300, acc = 0;
    uint32_t wa = 0, wb = 0;
    float cur[32];
    for (int k = 0; k < 32; ++k) cur[k] = cst[k + threadIdx.x];
    for (int k = threadIdx.x; k < 4096; k += 64) lds[k] = k;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N / 32; ++it) {
        double e[32];
        if (MODE >= 3) {
#pragma unroll
            for (int k = 0; k < 32; ++k) e[k] = lds[(it * 32 + k) & 4095];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            double &up = (k & 1) ? u1 : u0;
            const double diag = (k & 1) ? u0 : u1;
            if (MODE >= 3) { up = e[k]; shift_in(up, g); } else shift_in(up, g);
            const double c = (double)cur[k];
            const double p1 = diag + c, p2 = g + c, p3 = up + c;
            const double m12 = __builtin_fmin(p1, p2);
            const double best = __builtin_fmin(m12, p3);
            if (MODE >= 1) { plane_bit(wa, p2, p1); plane_bit(wb, p3, m12); }
            else { wa = wa + wa + (uint32_t)(p2 < p1); wb = wb + wb + (uint32_t)(p3 < m12); }
            if (MODE >= 2) push_lane63(acc, best);
            g = best;
        }
        planes[it * 64 + threadIdx.x] = wa; planes[(it + 512) * 64 + threadIdx.x] = wb;
        if (MODE >= 2 && threadIdx.x < 32) lds[(it * 32 + 31 - threadIdx.x) & 2047] = acc;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = g + u0 + u1 + acc;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
template <int MODE> float run(double *out, long long *cyc, float *cst, uint32_t *pl) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, 1, 64, 0, 0, out, cyc, cst, pl);
    hipEventRecord(a); hipLaunchKernelGGL(probe<MODE>, 1, 64, 0, 0, out, cyc, cst, pl); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    double *out; long long *cyc; float *cst; uint32_t *pl;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8); hipMalloc(&cst, 4096); hipMalloc(&pl, 1024 * 64 * 4 * 2);
    hipMemset(cst, 0, 4096);
    float ms[4] = {run<0>(out, cyc, cst, pl), run<1>(out, cyc, cst, pl), run<2>(out, cyc, cst, pl), run<3>(out, cyc, cst, pl)};
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char *nm[] = {"C planes (baseline 56.6)", "asm cmp+addc planes", "+DPP shift-register publish", "+edge from preloaded regs"};
    for (int m = 0; m < 4; ++m) printf("%-32s %7.2f ticks/step (%.3f ms)\n", nm[m], (double)h[m] / N, ms[m]);
    return 0;
}
