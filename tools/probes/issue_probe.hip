// Issue cost of single instructions for ONE wave64 on gfx950 (cycles per instruction, independent operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define PROBE(NAME, ASM)                                                                              \
    __global__ void NAME(long long *cyc, double *out) {                                                  \
        __shared__ double lds[1024];                                                                     \
        double a = threadIdx.x, b = 2.0 + threadIdx.x, c = 0, d = 0, e = 0, f = 0;                       \
        float x = threadIdx.x; uint32_t w = threadIdx.x, w2 = 1; uint32_t addr = threadIdx.x * 8;       \
        lds[threadIdx.x] = 0;                                                                            \
        long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < 64; ++it) {                                                                \
            asm volatile(".rept " STR(REP) "\n\t" ASM "\n\t.endr"                                        \
                         : "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(w), "+v"(w2)                          \
                         : "v"(a), "v"(b), "v"(x), "v"(addr) : "vcc", "memory");                        \
        }                                                                                                \
        long long t1 = __builtin_readcyclecounter();                                                    \
        out[threadIdx.x] = c + d + e + f + w + w2 + lds[threadIdx.x];                                    \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                          \
    }
// operands: %0..%3 doubles (out), %4,%5 u32 (out), %6,%7 doubles (in), %8 float (in), %9 u32 lds address
PROBE(p_add_f64, "v_add_f64 %0, %6, %7")
PROBE(p_min_f64, "v_min_f64 %0, %6, %7")
PROBE(p_cmp_f64, "v_cmp_lt_f64 vcc, %6, %7")
PROBE(p_cmp_u64, "v_cmp_ne_u64 vcc, %6, %7")
PROBE(p_cmp_f32, "v_cmp_lt_f32 vcc, %8, %8")
PROBE(p_addc, "v_addc_co_u32 %4, vcc, %4, %4, vcc")
PROBE(p_cvt, "v_cvt_f64_f32 %0, %8")
PROBE(p_mov, "v_mov_b32 %4, %5")
PROBE(p_dpp, "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(p_dswr, "ds_write_b64 %9, %6")
PROBE(p_dsrd, "ds_read_b64 %0, %9")
PROBE(p_add_dep, "v_add_f64 %0, %0, %7")
PROBE(p_min_dep, "v_min_f64 %0, %0, %7")
PROBE(p_dpp_dep, "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(p_add_min, "v_add_f64 %0, %1, %7\n\tv_min_f64 %1, %0, %6")
PROBE(p_cmp_addc, "v_cmp_lt_f64 vcc, %6, %7\n\tv_addc_co_u32 %4, vcc, %4, %4, vcc")
PROBE(p_cndmask, "v_cndmask_b32 %4, %5, %4, vcc")
PROBE(p_lshl_or, "v_lshl_or_b32 %4, %4, 1, %5")
#define RUN(NAME, n) do { for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, cyc, out); hipDeviceSynchronize(); } \
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-12s %6.2f cycles/instr\n", #NAME, (double)h / (64.0 * REP * n)); } while (0)
int main() {
    long long *cyc; double *out; hipMalloc(&cyc, 8); hipMalloc(&out, 64 * 8);
    RUN(p_add_f64, 1); RUN(p_min_f64, 1); RUN(p_cmp_f64, 1); RUN(p_cmp_u64, 1); RUN(p_cmp_f32, 1); RUN(p_addc, 1); RUN(p_cvt, 1);
    RUN(p_mov, 1); RUN(p_dpp, 1); RUN(p_dswr, 1); RUN(p_dsrd, 1); RUN(p_add_dep, 1); RUN(p_min_dep, 1); RUN(p_dpp_dep, 1);
    RUN(p_add_min, 2); RUN(p_cmp_addc, 2); RUN(p_cndmask, 1); RUN(p_lshl_or, 1);
    return 0;
}
