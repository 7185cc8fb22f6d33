// Micro-probe 3: what does each part of one DTW sweep step cost for a single wave?  (cycles per step, s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N 4096
__device__ __forceinline__ void shift_in(double &up, double g) {
    union { double d; int i[2]; } s, o; s.d = g; o.d = up;
    o.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], 0x138, 0xf, 0xf, false);
    o.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], 0x138, 0xf, 0xf, false);
    up = o.d;
}
__device__ __forceinline__ void plane_bit(uint32_t &w, double a, double b) {
    asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(a), "v"(b) : "vcc");
}
// MODE bits: 1 = cvt from float (else doubles preloaded), 2 = DPP shift, 4 = plane bits, 8 = p1/p3 adds + second min,
// 16 = ds_write publish
template <int MODE>
__global__ void probe(double *out, long long *cyc, const float *cst, uint32_t *planes) {
    __shared__ double lds[4096];
    double g = threadIdx.x * 1e-3, u0 = 1e300, u1 = 1e300;
    uint32_t wa = 0, wb = 0;
    float cur[32];
    double curd[32];
    for (int k = 0; k < 32; ++k) { cur[k] = cst[k + threadIdx.x]; curd[k] = cur[k]; }
    double *pub = lds + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < N / 32; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            double &up = (k & 1) ? u1 : u0;
            const double diag = (k & 1) ? u0 : u1;
            if (MODE & 2) shift_in(up, g);
            const double c = (MODE & 1) ? (double)cur[k] : curd[k];
            const double p2 = g + c;
            double best, m12 = p2, p1 = p2, p3 = p2;
            if (MODE & 8) {
                p1 = diag + c;
                p3 = up + c;
                m12 = __builtin_fmin(p1, p2);
                best = __builtin_fmin(m12, p3);
            } else {
                best = __builtin_fmin(p2, up);
            }
            if (MODE & 4) { plane_bit(wa, p2, p1); plane_bit(wb, p3, m12); }
            g = best;
            if (MODE & 16) pub[k] = best;
        }
        if (MODE & 4) { planes[it * 64 + threadIdx.x] = wa ^ wb; }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = g + u0 + u1 + lds[(threadIdx.x * 7) & 4095];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, double *out, long long *cyc, float *cst, uint32_t *pl) {
    long long h = 0;
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, cst, pl); hipDeviceSynchronize(); }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s %6.1f cycles/step\n", name, (double)h / N);
}
int main() {
    double *out; long long *cyc; float *cst; uint32_t *pl;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8); hipMalloc(&cst, 4096 * 4); hipMalloc(&pl, 64 * N * 4);
    hipMemset(cst, 0, 4096 * 4);
    run<0>("add+min (chain only, f64 preloaded)", out, cyc, cst, pl);
    run<1>("+ cvt", out, cyc, cst, pl);
    run<2>("chain + dpp", out, cyc, cst, pl);
    run<2 | 8>("chain + dpp + p1,p3,min", out, cyc, cst, pl);
    run<1 | 2 | 8>("+ cvt", out, cyc, cst, pl);
    run<2 | 4 | 8>("chain + dpp + p1,p3,min + planes (no cvt)", out, cyc, cst, pl);
    run<1 | 2 | 4 | 8>("full step", out, cyc, cst, pl);
    run<1 | 2 | 4 | 8 | 16>("full step + ds_write", out, cyc, cst, pl);
    return 0;
}
