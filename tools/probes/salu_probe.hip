// Issue / dependent-issue cost of scalar instructions for ONE wave on gfx950 (cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define PROBE(NAME, ASM)                                                                                  \
    __global__ void NAME(long long *cyc, int *out, int seed) {                                            \
        int a = seed, b = seed + 1, c = seed + 2; int v = threadIdx.x; unsigned long long q = seed;       \
        long long t0 = __builtin_readcyclecounter();                                                    \
        for (int it = 0; it < 64; ++it) {                                                                \
            asm volatile(".rept " STR(REP) "\n\t" ASM "\n\t.endr"                                        \
                         : "+s"(a), "+s"(b), "+s"(c), "+v"(v), "+s"(q) : : "vcc", "scc", "memory");    \
        }                                                                                                \
        long long t1 = __builtin_readcyclecounter();                                                    \
        out[threadIdx.x] = a + b + c + v + (int)q;                                                       \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                          \
    }
PROBE(s_add_indep, "s_add_i32 %0, %1, %2")
PROBE(s_add_dep, "s_add_i32 %0, %0, %1")
PROBE(s_add_dep2, "s_add_i32 %0, %0, %1\n\ts_add_i32 %2, %2, %1")
PROBE(s_lshl64_dep, "s_lshl_b64 %4, %4, 1")
PROBE(s_ff1_dep, "s_ff1_i32_b64 %0, %4\n\ts_lshr_b64 %4, %4, %0")
PROBE(s_cmp_csel, "s_cmp_lt_i32 %0, %1\n\ts_cselect_b32 %0, %1, %2")
PROBE(rdlane_dep, "v_readlane_b32 %0, %3, 5\n\ts_add_i32 %1, %1, %0")
PROBE(rdlane_sel, "s_and_b32 %1, %1, 63\n\tv_readlane_b32 %0, %3, %1\n\ts_add_i32 %1, %1, %0")
PROBE(v_then_s, "v_add_u32 %3, %3, %3\n\ts_add_i32 %0, %0, %1")
PROBE(vcmp_cnd, "v_cmp_eq_u32 vcc, %0, %3\n\tv_cndmask_b32 %3, %3, %3, vcc")
PROBE(branch_taken, "s_branch 1f\n\ts_nop 0\n1:")
PROBE(branch_cond_nt, "s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc0 1f\n1:")
#define RUN(NAME, n) do { for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(NAME, dim3(1), dim3(64), 0, 0, cyc, out, 3); hipDeviceSynchronize(); } \
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-16s %6.2f cycles/instr (%d instr/rep)\n", #NAME, (double)h / (64.0 * REP * n), n); } while (0)
int main() {
    long long *cyc; int *out; hipMalloc(&cyc, 8); hipMalloc(&out, 64 * 4);
    RUN(s_add_indep, 1); RUN(s_add_dep, 1); RUN(s_add_dep2, 2); RUN(s_lshl64_dep, 1); RUN(s_ff1_dep, 2); RUN(s_cmp_csel, 2);
    RUN(rdlane_dep, 2); RUN(rdlane_sel, 3); RUN(v_then_s, 2); RUN(vcmp_cnd, 2); RUN(branch_taken, 1); RUN(branch_cond_nt, 2);
    return 0;
}
