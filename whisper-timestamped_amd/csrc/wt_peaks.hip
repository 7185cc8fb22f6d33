// Disfluency detection on the local-cost matrix: where, inside the frames a token owns, does its LAST attention
// peak start?
//
// Replaces /root/reference/whisper_timestamped/transcribe.py:1656-1672: for every token row t with frames
// [jumps[t], jumps[t+1]) the reference runs
//     peaks, properties = scipy.signal.find_peaks(-cost[t, begin:end], width=3, prominence=0.02)
// and, when more than one peak survives, moves the token's start to round(properties["left_ips"][-1]) + begin.
// scipy's algorithm (scipy/signal/_peak_finding_utils.pyx: _local_maxima_1d, _peak_prominences, _peak_widths; an
// installed dependency of the reference, restated here operation by operation, all in f64 like scipy):
//   * a local maximum is a sample -- or the middle of a plateau -- strictly above both neighbours; the first and
//     the last sample never are;
//   * its prominence is its height above the higher of the two lowest points met while walking left / right until a
//     strictly higher sample (or the end); those lowest points are its bases (among equal lows the one met first,
//     i.e. the one nearest to the peak);
//   * kept if prominence >= 0.02; its width is measured at half prominence, between the two crossings (linear
//     interpolation between samples) searched from the peak towards its bases; kept if width >= 3.
// The matrix is the fp32 cost the cost kernel wrote (the reference's f64 matrix holds exactly these values), the
// token spans are the jumps the DTW kernel wrote: nothing crosses PCIe but the T+1 integers of the answer.
//
// Mapping: one workgroup per unit, one THREAD per token row (spans are a few frames; the rows are independent).
#include "wt_common.h"

namespace wt {

__global__ __launch_bounds__(256) void disfluency_kernel(const float *__restrict__ cost, const wt_seg_desc *__restrict__ segs,
                                                         const int32_t *__restrict__ jumps, int32_t *__restrict__ jumps_start,
                                                         double min_prominence, double min_width) {
    const wt_seg_desc d = segs[blockIdx.x];
    const int T = d.T, F = d.F;
    const int32_t *jp = jumps + d.jumps_offset;
    int32_t *js = jumps_start + d.jumps_offset;
    for (int t = threadIdx.x; t <= T; t += blockDim.x) {
        if (t == T) {
            js[T] = jp[T];
            continue;
        }
        const int begin = jp[t], n = jp[t + 1] - begin;
        const float *row = cost + d.cost_offset + (int64_t)t * F + begin;
        auto x = [&](int i) { return -(double)row[i]; };   // the attention profile of the token (>= 0)
        int kept = 0;
        double last_left = 0.0;
        const int i_last = n - 1;
        for (int i = 1; i < i_last; ++i) {
            const double xi = x(i);
            if (!(x(i - 1) < xi)) continue;
            int ahead = i + 1;
            while (ahead < i_last && x(ahead) == xi) ++ahead;   // plateau
            if (!(x(ahead) < xi)) continue;
            const int peak = (i + ahead - 1) / 2;
            i = ahead;                                          // (the loop's ++i follows, as in scipy)
            // prominence and bases
            double left_min = xi, right_min = xi;
            int lb = peak, rb = peak;
            for (int k = peak; k >= 0; --k) {
                const double v = x(k);
                if (!(v <= xi)) break;
                if (v < left_min) { left_min = v; lb = k; }
            }
            for (int k = peak; k <= i_last; ++k) {
                const double v = x(k);
                if (!(v <= xi)) break;
                if (v < right_min) { right_min = v; rb = k; }
            }
            const double prominence = xi - fmax(left_min, right_min);
            if (!(min_prominence <= prominence)) continue;
            // width at half prominence
            const double height = xi - prominence * 0.5;
            int k = peak;
            while (lb < k && height < x(k)) --k;
            double left_ip = (double)k;
            if (x(k) < height) left_ip += (height - x(k)) / (x(k + 1) - x(k));
            k = peak;
            while (k < rb && height < x(k)) ++k;
            double right_ip = (double)k;
            if (x(k) < height) right_ip -= (height - x(k)) / (x(k - 1) - x(k));
            if (!(min_width <= right_ip - left_ip)) continue;
            ++kept;
            last_left = left_ip;
        }
        js[t] = kept > 1 ? (int)__builtin_rint(last_left) + begin : begin;   // Python's round(): half to even
    }
}

int disfluency_batch(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int32_t *jumps, int32_t *jumps_start,
                     double min_prominence, double min_width, hipStream_t st) {
    if (!cost || !segs_dev || !jumps || !jumps_start || n_seg < 0) {
        set_error("wt_disfluency_batch: null pointer or bad count");
        return WT_E_BADARG;
    }
    if (n_seg == 0) return WT_OK;
    hipLaunchKernelGGL(disfluency_kernel, dim3(n_seg), dim3(256), 0, st, cost, segs_dev, jumps, jumps_start, min_prominence,
                       min_width);
    WT_HIP(hipGetLastError());
    return WT_OK;
}

}  // namespace wt
