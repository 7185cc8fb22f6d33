// extern "C" surface of libwtalign.so (see include/wtalign.h) + host plumbing.
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>

#include "wt_common.h"
#include "wt_small.h"

namespace wt {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return WT_E_HIP;
}

struct Arena {
    void *p = nullptr;
    size_t cap = 0;
    const void *tag_ptr = nullptr;  // what the owner last prepared in this arena (e.g. which filterbank)
    long long tag_val = 0;
};
static std::mutex g_mu;
// One arena per (device, purpose, stream): calls on distinct streams never share scratch (they may run concurrently),
// calls on one stream are ordered by the stream.
static std::map<std::tuple<int, int, hipStream_t>, Arena> g_arena;

// Growing is a hipMalloc (synchronising, not graph-capturable); the first call reserves at least 256 KB so that
// steady state never grows.  `fresh` (optional) tells the caller that the memory is new (tags were reset).
static int scratch_ex(int purpose, hipStream_t st, size_t bytes, void **out, Arena **arena) {
    int dev = 0;
    WT_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    Arena &a = g_arena[std::make_tuple(dev, purpose, st)];
    if (bytes > a.cap) {
        size_t want = bytes < (256u << 10) ? (256u << 10) : bytes * 2;
        void *np = nullptr;
        WT_HIP(hipMalloc(&np, want));
        if (a.p) {
            WT_HIP(hipDeviceSynchronize());
            WT_HIP(hipFree(a.p));
        }
        a.p = np;
        a.cap = want;
        a.tag_ptr = nullptr;
        a.tag_val = 0;
    }
    *out = a.p;
    if (arena) *arena = &a;
    return WT_OK;
}
int scratch(hipStream_t st, size_t bytes, void **out) { return scratch_ex(0, st, bytes, out, nullptr); }  // cost path
int scratch_dtw(hipStream_t st, size_t bytes, void **out) { return scratch_ex(2, st, bytes, out, nullptr); }  // DTW planes
// log-mel path: returns true in *prepared when the arena already holds the preparation tagged (ptr, val)
int scratch_tagged(hipStream_t st, size_t bytes, void **out, const void *tag_ptr, long long tag_val, bool *prepared) {
    Arena *a = nullptr;
    int rc = scratch_ex(1, st, bytes, out, &a);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g_mu);
    *prepared = (a->tag_ptr == tag_ptr && a->tag_val == tag_val && tag_ptr != nullptr);
    a->tag_ptr = tag_ptr;
    a->tag_val = tag_val;
    return WT_OK;
}
void scratch_forget_tags() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_arena) kv.second.tag_ptr = nullptr;
}

int cost_batch(const void *, int, const wt_seg_desc *, const wt_seg_desc *, int, const int32_t *, int, int, float, float *,
               bool, bool, hipStream_t);
int dtw_batch(const float *, const wt_seg_desc *, const wt_seg_desc *, int, int, int32_t *, int32_t *, int32_t *, int32_t *,
              double *, bool, hipStream_t);
int align_small_tail(const wt_seg_desc *, const wt_seg_desc *, int, float *, bool, int32_t *, int32_t *, int32_t *, int32_t *,
                     double *, hipStream_t);
int logprob_gather_batch(const void *, int, int64_t, int, int, const int32_t *, const uint8_t *, int, const int32_t *, float *,
                         hipStream_t);
int logprob_digest_streams(const float *, int64_t, int, int, const void *, int, int64_t, const int32_t *, int64_t, int64_t,
                           const int32_t *, int, int, float *, float *, hipStream_t);
int qk_rows_batch(const void *const *, const void *const *, int, int, int, int, int64_t, int64_t, int, int, int, float,
                  const int32_t *, const int32_t *, const int32_t *, int, const int32_t *, const int32_t *, const int32_t *, void *,
                  int, int64_t, int64_t, int64_t, hipStream_t);
int find_start_padding_batch(const float *, int, int, int, int32_t *, hipStream_t);
int disfluency_batch(const float *, const wt_seg_desc *, int, const int32_t *, int32_t *, double, double, hipStream_t);
int logmel_batch(const float *, int, int64_t, const int32_t *, const float *, int, int, float *, float *, int32_t *, hipStream_t);
int capture_rows(const void *, int, int, int, int, const int32_t *, const int32_t *, int, void *, int, int64_t, int64_t,
                 hipStream_t);
int qk_rows(const void *, const void *, int, int, int, int, int, float, const int32_t *, const int32_t *, int, void *, int,
            int64_t, int64_t, hipStream_t);

}  // namespace wt

extern "C" {

int wt_version(void) { return WT_ABI_VERSION; }

const char *wt_last_error(void) { return wt::g_err; }

int wt_shutdown(void) {
    std::lock_guard<std::mutex> lk(wt::g_mu);
    for (auto &kv : wt::g_arena) {
        if (kv.second.p) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            (void)hipSetDevice(std::get<0>(kv.first));
            (void)hipFree(kv.second.p);
            (void)hipSetDevice(cur);
        }
    }
    wt::g_arena.clear();
    return WT_OK;
}

// Arenas are keyed by (device, purpose, stream): without this a process that keeps creating streams would keep one arena
// set per stream until wt_shutdown.  The caller guarantees the stream's work has completed.
int wt_release_stream(void *stream) {
    std::lock_guard<std::mutex> lk(wt::g_mu);
    int freed = 0, cur = 0;
    (void)hipGetDevice(&cur);
    for (auto it = wt::g_arena.begin(); it != wt::g_arena.end();) {
        if (std::get<2>(it->first) == (hipStream_t)stream) {
            if (it->second.p) {
                (void)hipSetDevice(std::get<0>(it->first));
                (void)hipFree(it->second.p);
                ++freed;
            }
            it = wt::g_arena.erase(it);
        } else
            ++it;
    }
    (void)hipSetDevice(cur);
    return freed;
}

int wt_capture_rows(const void *qk, int qk_dtype, int n_heads, int n_q, int n_ctx, const int32_t *heads, const int32_t *slots,
                    int n_sel, void *ring, int ring_dtype, int64_t ring_rows, int64_t row, void *stream) {
    return wt::capture_rows(qk, qk_dtype, n_heads, n_q, n_ctx, heads, slots, n_sel, ring, ring_dtype, ring_rows, row,
                            (hipStream_t)stream);
}

int wt_qk_rows(const void *q, const void *k, int dtype, int n_rows, int n_ctx, int d_model, int head_dim, float scale,
               const int32_t *heads, const int32_t *slots, int n_sel, void *ring, int ring_dtype, int64_t ring_rows,
               int64_t row0, void *stream) {
    return wt::qk_rows(q, k, dtype, n_rows, n_ctx, d_model, head_dim, scale, heads, slots, n_sel, ring, ring_dtype, ring_rows,
                       row0, (hipStream_t)stream);
}

int wt_cost_batch(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                  const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, void *stream) {
    return wt::cost_batch(qk, qk_dtype, segs_host, segs_dev, n_seg, head_idx, n_heads, medfilt_width, qk_scale, cost, false, false,
                          (hipStream_t)stream);
}

int wt_dtw_batch(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, int32_t *jumps,
                 int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, void *stream) {
    return wt::dtw_batch(cost, segs_host, segs_dev, n_seg, WT_STEP_SYMMETRIC1, jumps, path_i, path_j, path_len, dist, false,
                         (hipStream_t)stream);
}

int wt_dtw_batch_pattern(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                         int step_pattern, int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist,
                         void *stream) {
    return wt::dtw_batch(cost, segs_host, segs_dev, n_seg, step_pattern, jumps, path_i, path_j, path_len, dist, false,
                         (hipStream_t)stream);
}

// rowmean for every unit; then the units that qualify (wt_small.h: the LDS their matrix, planes and boundary rows need --
// any T <= 256 whose (T, F) fits 160 KB; a property of the unit's own shape) take the fused tail kernel -- column norm, cost[0,0], DTW and backtrack in one workgroup -- and the
// others colnorm / fix00 / dtw, which skip the small ones.
int wt_align_batch_v3(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                      const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, int32_t *jumps,
                      int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, int flags, void *stream) {
    const bool fused = !(flags & WT_ALIGN_NO_FUSED_SMALL_UNITS);
    bool any_small = false;
    if (fused && segs_host && n_seg > 0)
        for (int s = 0; s < n_seg && !any_small; ++s) any_small = wt::wt_small_unit(segs_host[s].T, segs_host[s].F);
    if (!jumps || (!path_i != !path_j)) {   // (before anything is launched; the batched entry points check the rest)
        wt::set_error("wt_align_batch: null pointer or bad count");
        return WT_E_BADARG;
    }
    int rc = wt::cost_batch(qk, qk_dtype, segs_host, segs_dev, n_seg, head_idx, n_heads, medfilt_width, qk_scale, cost, any_small,
                            (flags & WT_ALIGN_ROWS_PER_CLASS) != 0, (hipStream_t)stream);
    if (rc) return rc;
    if (any_small) {
        rc = wt::align_small_tail(segs_host, segs_dev, n_seg, cost, (flags & WT_ALIGN_KEEP_COST) != 0, jumps, path_i, path_j,
                                  path_len, dist, (hipStream_t)stream);
        if (rc) return rc;
    }
    return wt::dtw_batch(cost, segs_host, segs_dev, n_seg, WT_STEP_SYMMETRIC1, jumps, path_i, path_j, path_len, dist, any_small,
                         (hipStream_t)stream);
}

int wt_align_batch(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                   const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, int32_t *jumps,
                   int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, void *stream) {
    return wt_align_batch_v3(qk, qk_dtype, segs_host, segs_dev, n_seg, head_idx, n_heads, medfilt_width, qk_scale, cost, jumps,
                             path_i, path_j, path_len, dist, WT_ALIGN_KEEP_COST, stream);
}

int wt_disfluency_batch(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int32_t *jumps, int32_t *jumps_start,
                        double min_prominence, double min_width, void *stream) {
    return wt::disfluency_batch(cost, segs_dev, n_seg, jumps, jumps_start, min_prominence, min_width, (hipStream_t)stream);
}

int wt_find_start_padding_batch(const float *mel, int n_chunks, int n_mels, int n_cols, int32_t *out, void *stream) {
    return wt::find_start_padding_batch(mel, n_chunks, n_mels, n_cols, out, (hipStream_t)stream);
}

int wt_logprob_gather_batch(const void *logits, int logits_dtype, int64_t row_stride, int n_rows, int V,
                            const int32_t *token, const uint8_t *suppress, int suppress_rows, float *out, void *stream) {
    return wt::logprob_gather_batch(logits, logits_dtype, row_stride, n_rows, V, token, suppress, suppress_rows, nullptr, out,
                                    (hipStream_t)stream);
}

int wt_logprob_gather_rows(const void *logits, int logits_dtype, int64_t row_stride, const int32_t *row_index, int n_out,
                           int V, const int32_t *token, float *out, void *stream) {
    if (!row_index) {
        wt::set_error("wt_logprob_gather_rows: row_index is null");
        return WT_E_BADARG;
    }
    return wt::logprob_gather_batch(logits, logits_dtype, row_stride, n_out, V, token, nullptr, 0, row_index, out,
                                    (hipStream_t)stream);
}

int wt_logprob_digest_streams(const float *logits, int64_t row_stride, int n_rows, int V, const void *token, int token_dtype,
                              int64_t token_stride, const int32_t *ring_index, int64_t ring_rows, int64_t ring_row,
                              const int32_t *aux_tokens_host, int n_aux, int slice_begin, float *digest, float *slice,
                              void *stream) {
    return wt::logprob_digest_streams(logits, row_stride, n_rows, V, token, token_dtype, token_stride, ring_index, ring_rows,
                                      ring_row, aux_tokens_host, n_aux, slice_begin, digest, slice, (hipStream_t)stream);
}

int wt_qk_rows_batch(const void *const *q_layers_host, const void *const *k_layers_host, int n_layers, int dtype, int n_batch,
                     int n_q, int64_t q_batch_stride, int64_t k_batch_stride, int n_ctx, int d_model, int head_dim,
                     float scale, const int32_t *sel_layer, const int32_t *sel_head, const int32_t *sel_slot, int n_sel,
                     const int32_t *row_begin, const int32_t *row_end, void *ring, int ring_dtype, int64_t ring_batch_stride,
                     int64_t ring_rows, int64_t ring_row0, void *stream) {
    return wt::qk_rows_batch(q_layers_host, k_layers_host, n_layers, dtype, n_batch, n_q, q_batch_stride, k_batch_stride, n_ctx,
                             d_model, head_dim, scale, sel_layer, sel_head, sel_slot, n_sel, row_begin, row_end, nullptr, ring,
                             ring_dtype, ring_batch_stride, ring_rows, ring_row0, (hipStream_t)stream);
}

int wt_qk_rows_streams(const void *const *q_layers_host, const void *const *k_layers_host, int n_layers, int dtype, int n_batch,
                       int n_q, int64_t q_batch_stride, int64_t k_batch_stride, int n_ctx, int d_model, int head_dim, float scale,
                       const int32_t *sel_layer, const int32_t *sel_head, const int32_t *sel_slot, int n_sel,
                       const int32_t *ring_index, void *ring, int ring_dtype, int64_t ring_batch_stride, int64_t ring_rows,
                       int64_t ring_row0, void *stream) {
    if (!ring_index) {
        wt::set_error("wt_qk_rows_streams: ring_index is null");
        return WT_E_BADARG;
    }
    return wt::qk_rows_batch(q_layers_host, k_layers_host, n_layers, dtype, n_batch, n_q, q_batch_stride, k_batch_stride, n_ctx,
                             d_model, head_dim, scale, sel_layer, sel_head, sel_slot, n_sel, nullptr, nullptr, ring_index, ring,
                             ring_dtype, ring_batch_stride, ring_rows, ring_row0, (hipStream_t)stream);
}

int wt_logmel_batch(const float *pcm, int n_chunks, int64_t n_samples, const int32_t *n_valid_samples, const float *mel_fb,
                    int n_mels, int n_frames, float *mel_out, float *gmax, void *stream) {
    return wt::logmel_batch(pcm, n_chunks, n_samples, n_valid_samples, mel_fb, n_mels, n_frames, mel_out, gmax, nullptr,
                            (hipStream_t)stream);
}

int wt_logmel_pad_batch(const float *pcm, int n_chunks, int64_t n_samples, const int32_t *n_valid_samples, const float *mel_fb,
                        int n_mels, int n_frames, float *mel_out, float *gmax, int32_t *start_of_padding, void *stream) {
    if (!start_of_padding) {
        wt::set_error("wt_logmel_pad_batch: start_of_padding is null");
        return WT_E_BADARG;
    }
    return wt::logmel_batch(pcm, n_chunks, n_samples, n_valid_samples, mel_fb, n_mels, n_frames, mel_out, gmax, start_of_padding,
                            (hipStream_t)stream);
}

}  // extern "C"
