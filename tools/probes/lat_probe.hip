// Micro-probe: dependent-chain latency of the instructions on the DTW critical path (one wave, gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
__device__ __forceinline__ double shr1(double v, double o) {
    union { double d; int i[2]; } s, r, oo; s.d = v; oo.d = o;
    r.i[0] = __builtin_amdgcn_update_dpp(oo.i[0], s.i[0], 0x138, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(oo.i[1], s.i[1], 0x138, 0xf, 0xf, false);
    return r.d;
}
template <int MODE>
__global__ void probe(double *out, long long *cyc, double c) {
    double g = threadIdx.x * 1e-3, h = g + 1.0;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N / 16; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (MODE == 0) g = g + c;                                   // v_add_f64 chain
            if (MODE == 1) g = __builtin_fmin(g + c, h);                // add + min
            if (MODE == 2) g = shr1(g, c) + c;                          // dpp + add
            if (MODE == 3) { double up = shr1(g, c); g = __builtin_fmin(__builtin_fmin(h + c, g + c), up + c); } // full step chain
            if (MODE == 4) { float f = (float)g; f = f + (float)c; g = f; }  // f32 add via cvt (ignore)
            if (MODE == 5) { g = g + c; h = h + c; }                    // two independent f64 chains
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = g + h;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
    hipLaunchKernelGGL(probe<0>, 1, 64, 0, 0, out, cyc, 1e-9);
    hipLaunchKernelGGL(probe<1>, 1, 64, 0, 0, out, cyc, 1e-9);
    hipLaunchKernelGGL(probe<2>, 1, 64, 0, 0, out, cyc, 1e-9);
    hipLaunchKernelGGL(probe<3>, 1, 64, 0, 0, out, cyc, 1e-9);
    hipLaunchKernelGGL(probe<5>, 1, 64, 0, 0, out, cyc, 1e-9);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char *nm[] = {"add_f64 chain", "add+min chain", "dpp+add chain", "dtw step chain", "", "2 indep add chains"};
    for (int m : {0, 1, 2, 3, 5}) printf("%-20s %8.2f s_memtime ticks per iteration\n", nm[m], (double)h[m] / N);
    return 0;
}
