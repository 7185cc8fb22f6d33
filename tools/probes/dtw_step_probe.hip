// Micro-probe: cost of one DTW sweep step (one wave, registers only) as pieces are added; and shader clock.
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
template <int MODE>
__global__ void probe(double *out, long long *cyc, const float *cst, uint32_t *planes) {
    __shared__ double lds[128];
    double g = threadIdx.x * 1e-3, u0 = 1e300, u1 = 1e300;
    uint32_t wa = 0, wb = 0;
    float cur[32];
    for (int k = 0; k < 32; ++k) cur[k] = cst[k + threadIdx.x];
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N / 32; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            double &up = (k & 1) ? u1 : u0;
            const double diag = (k & 1) ? u0 : u1;
            shift_in(up, g);
            const double c = (MODE >= 1) ? (double)cur[k] : 1e-9;
            const double p1 = diag + c, p2 = g + c, p3 = up + c;
            const double m12 = __builtin_fmin(p1, p2);
            const double best = __builtin_fmin(m12, p3);
            if (MODE >= 2) { wa = wa + wa + (uint32_t)(p2 < p1); wb = wb + wb + (uint32_t)(p3 < m12); }
            if (MODE == 3 && threadIdx.x == 63) lds[k] = best;
            if (MODE == 4) lds[k] = best;
            g = best;
        }
        if (MODE >= 2) { planes[it * 64 + threadIdx.x] = wa; planes[(it + 512) * 64 + threadIdx.x] = wb; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = g + u0 + u1 + lds[threadIdx.x & 31];
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
    float ms[5] = {run<0>(out, cyc, cst, pl), run<1>(out, cyc, cst, pl), run<2>(out, cyc, cst, pl), run<3>(out, cyc, cst, pl),
                   run<4>(out, cyc, cst, pl)};
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char *nm[] = {"chain only", "+cvt cost", "+2 cmp + bit planes", "+lane63 ds_write", "+all-lane same-addr ds_write"};
    for (int m = 0; m < 5; ++m)
        printf("%-30s %7.2f ticks/step   kernel %.3f ms -> %.1f ticks/us\n", nm[m], (double)h[m] / N, ms[m], h[m] / (ms[m] * 1e3));
    return 0;
}
