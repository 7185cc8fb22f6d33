// Probe: what read-only HBM bandwidth is achievable on this box (ceiling for the log-prob gather stream)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const float4 *__restrict__ x, size_t n4, float *out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 r[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NT) { f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(x + i + u * stride)); r[u] = make_float4(t.x, t.y, t.z, t.w); } else r[u] = x[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (r[u].x + r[u].y) + (r[u].z + r[u].w);
    }
    for (; i < n4; i += stride) { float4 r = x[i]; acc += (r.x + r.y) + (r.z + r.w); }
    if (acc == 1.2345f) out[0] = acc;
}
// block-per-row like logprob_gather (rows of V floats)
__global__ __launch_bounds__(256) void rd_rows(const float *__restrict__ x, int V, float *out) {
    const float4 *r = reinterpret_cast<const float4 *>(x + (size_t)blockIdx.x * V);  // rows not 16B aligned in general: probe only
    float acc = 0.f;
    for (int v = threadIdx.x; v < V / 4; v += 256) { float4 q = r[v]; acc += (q.x + q.y) + (q.z + q.w); }
    if (acc == 1.2345f) out[0] = acc;
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const size_t bytes = (size_t)7168 * 51864 * 4;  // ~1.49 GB, the bench's logits
    float *x, *out; hipMalloc(&x, bytes); hipMalloc(&out, 4); hipMemset(x, 0, bytes);
    const size_t n4 = bytes / 16;
    for (int g : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32, 7168}) {
        float m1 = timeit([&] { hipLaunchKernelGGL((rd<1, false>), dim3(g), dim3(256), 0, 0, (const float4 *)x, n4, out); });
        float m4 = timeit([&] { hipLaunchKernelGGL((rd<4, false>), dim3(g), dim3(256), 0, 0, (const float4 *)x, n4, out); });
        float m8 = timeit([&] { hipLaunchKernelGGL((rd<8, false>), dim3(g), dim3(256), 0, 0, (const float4 *)x, n4, out); });
        float n4t = timeit([&] { hipLaunchKernelGGL((rd<4, true>), dim3(g), dim3(256), 0, 0, (const float4 *)x, n4, out); });
        printf("grid %6d: unroll1 %.0f GB/s  unroll4 %.0f  unroll8 %.0f  unroll4+nontemporal %.0f\n", g, bytes / m1 / 1e6, bytes / m4 / 1e6,
               bytes / m8 / 1e6, bytes / n4t / 1e6);
    }
    float mr = timeit([&] { hipLaunchKernelGGL(rd_rows, dim3(7168), dim3(256), 0, 0, x, 51864, out); });
    printf("block-per-row (7168 rows x 51864): %.0f GB/s\n", bytes / mr / 1e6);
    return 0;
}
