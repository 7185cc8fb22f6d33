// Probe: where does one K-full DTW unit spend its cycles?  Includes the product kernel source with WT_PROBE.
#define WT_PROBE 1
#include "../../whisper-timestamped_amd/csrc/wt_dtw.hip"
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <algorithm>
namespace wt { void set_error(const char *, ...) {} int hip_fail(hipError_t e, const char *w) { printf("HIP fail %s\n", w); return -2; }
int scratch(size_t, void **) { return 0; } }
int main(int argc, char **argv) {
    const int n = 32, T = argc > 1 ? atoi(argv[1]) : 224, F = argc > 2 ? atoi(argv[2]) : 1500;
    std::vector<wt_seg_desc> d(n);
    size_t per = ((size_t)T * F + 3) & ~3;
    for (int b = 0; b < n; ++b) { d[b] = {}; d[b].T = T; d[b].F = F; d[b].cost_offset = b * per; d[b].jumps_offset = b * (T + 1); d[b].pad_from = -1; }
    std::vector<float> c(n * per + 4);
    srand(1); for (auto &v : c) v = -(float)rand() / RAND_MAX;
    if (argc > 3 && atoi(argv[3]) == 1) {  // bench-like cost: small noise + a deep ridge on a random monotone staircase
        for (int b = 0; b < n; ++b) {
            std::vector<int> st(T);
            for (int t = 0; t < T; ++t) st[t] = rand() % F;
            std::sort(st.begin(), st.end());
            for (int t = 0; t < T; ++t)
                for (int f = 0; f < F; ++f) {
                    float v = -0.002f * ((float)rand() / RAND_MAX);
                    if (abs(f - st[t]) <= 1) v -= 0.3f;
                    c[b * per + (size_t)t * F + f] = v;
                }
        }
    }
    float *dc; wt_seg_desc *dd; int32_t *dj;
    hipMalloc(&dc, c.size() * 4); hipMalloc(&dd, n * sizeof(wt_seg_desc)); hipMalloc(&dj, n * (T + 1) * 4);
    hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), n * sizeof(wt_seg_desc), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        int rc = wt::dtw_batch(dc, d.data(), dd, n, dj, nullptr, nullptr, nullptr, nullptr, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long clk[16]; hipMemcpyFromSymbol(clk, HIP_SYMBOL(wt::wt_probe_clk), sizeof(clk));
        printf("rc=%d kernel %.1f us | wave start..end-of-forward (cycles from wave0 start):", rc, ms * 1e3);
        for (int w = 0; w < 4; ++w) printf("  w%d %lld..%lld", w, clk[w] - clk[0], clk[4 + w] - clk[0]);
        printf(" | backtrack %lld..%lld\n", clk[8] - clk[0], clk[9] - clk[0]);
    }
    return 0;
}
