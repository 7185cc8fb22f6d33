/*
 * wtalign.h -- C ABI of libwtalign.so: the MI355X (gfx950) word-alignment hot
 * path of whisper-timestamped.
 *
 * The reference (linto-ai/whisper-timestamped v1.15.9) is one pure-Python
 * module; it has no FFI for this path.  Each entry point below replaces the
 * numerics the reference reaches through Python at the cited lines of
 * /root/reference/whisper_timestamped/transcribe.py ("T.py:N" below), and is
 * what a ctypes binding inside the reference would call (INTEGRATION.md shows
 * that binding).
 *
 * Conventions
 *  - Every data pointer is a DEVICE pointer owned by the caller (PyTorch
 *    allocations via tensor.data_ptr()), except arguments named *_host.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *    calls are asynchronous with respect to that stream; nothing synchronises.
 *  - Return value: 0 on success; <0 on error (WT_E_*), with a message
 *    retrievable through wt_last_error() (thread-local).  Nothing throws.
 *  - The library keeps lazily-allocated device scratch arenas per (device,
 *    stream): per-unit reduction words, the banded mel filterbank, and the DTW's
 *    direction bit planes (T*F/4 bytes per unit of the largest batch seen);
 *    wt_shutdown() frees them.  Growing an arena is a hipMalloc: warm a stream
 *    up once before capturing calls on it into a HIP graph.
 */
#ifndef WTALIGN_H
#define WTALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WT_ABI_VERSION 5 /* 2: + wt_qk_rows_batch, wt_logprob_gather_rows, wt_dtw_batch_pattern; 3: + wt_align_batch_v3;
                            4: + wt_release_stream, wt_qk_rows_streams, wt_logmel_pad_batch; only the WT_API entries are exported (the library is built with
                               -fvisibility=hidden); 5: + wt_logprob_digest_streams */

/* The exported surface: exactly the functions marked WT_API below (tests/test_host_cpu.py holds `nm -D` to it). */
#define WT_API __attribute__((visibility("default")))

#define WT_OK 0
#define WT_E_BADARG (-1)      /* null pointer, negative size, bad dtype ...          */
#define WT_E_HIP (-2)         /* a HIP runtime call failed (see wt_last_error)       */
#define WT_E_UNSUPPORTED (-3) /* shape outside the kernels' range: T > 256 or F > 1792.
                                 The reference never exceeds T <= 226, F <= 1500        */

#define WT_DTYPE_F32 0
#define WT_DTYPE_F16 1

#define WT_N_AUDIO_CTX 1500 /* frames of 20 ms per 30 s window (T.py:44-47)       */
#define WT_MAX_TOKENS 256   /* rows of one DTW (decoder emits <= 224 + 2)         */
#define WT_MAX_FRAMES 1792
#define WT_MAX_LAYERS 32    /* hooked decoder layers of one wt_qk_rows_batch call     */

/* One alignment unit = one call of perform_word_alignment (T.py:1428): a
 * (heads, T tokens, F frames) window of cross-attention QK logits.
 * Element offsets (not bytes).  The QK logits of flat head k (= layer*H+head),
 * token row t, absolute frame f live at
 *     qk[qk_offset + k*head_stride + t*row_stride + f]
 * which covers both torch.cat(attention_weights) of T.py:1512 (head_stride =
 * T*1500, row_stride = 1500) and a device-side capture ring. */
typedef struct wt_seg_desc {
    int64_t qk_offset;
    int64_t head_stride;
    int64_t row_stride;
    int64_t cost_offset;  /* this unit's (T,F) row-major fp32 local-cost matrix in cost[]     */
    int64_t jumps_offset; /* this unit's T+1 int32 entries in jumps[]                        */
    int64_t path_offset;  /* this unit's (T+F-1)-capacity slot in path_i[] / path_j[]        */
    int32_t T;            /* tokens incl. the start/end timestamp tokens (T.py:1514)         */
    int32_t F;            /* end_token - start_token after T.py:1484-1489                    */
    int32_t start_token;  /* first absolute frame of the window (T.py:1540)                  */
    int32_t pad_from;     /* max_duration of T.py:1554-1565 when the mask applies, else -1 (0 is treated
                             like -1: the reference tests `if max_duration:`).
                             NB (reference quirk, reproduced): an ABSOLUTE frame index used
                             as a column index RELATIVE to start_token.                      */
} wt_seg_desc;

WT_API int wt_version(void);
WT_API const char *wt_last_error(void);
WT_API int wt_shutdown(void);

/* Frees the scratch arenas the library keeps for `stream` on every device (the arenas are keyed by (device, purpose,
 * stream): a process that creates and destroys streams calls this before hipStreamDestroy, otherwise one arena set
 * per stream stays allocated until wt_shutdown).  The stream's queued work must have completed (the call does not
 * synchronise it).  Returns the number of arenas freed (>= 0). */
WT_API int wt_release_stream(void *stream);

/* T.py:783-793 hook_attention_weights.  Copies the LAST query row of n_sel heads of one decoder layer's
 * cross-attention QK logits into the device capture ring (instead of the reference's `w[:, :, -1:, :].cpu()`
 * of all heads, once per token per layer).
 *   qk       : device, contiguous [n_heads][n_q][n_ctx] (the layer's (1,H,n_q,1500) tensor), qk_dtype
 *   heads    : device int32[n_sel], head index inside this layer
 *   slots    : device int32[n_sel], destination head slot in the ring
 *   ring     : device, [n_slots][ring_rows][n_ctx] of ring_dtype; writes ring[slots[i]][row][:]        */
WT_API int wt_capture_rows(const void *qk, int qk_dtype, int n_heads, int n_q, int n_ctx, const int32_t *heads,
                    const int32_t *slots, int n_sel, void *ring, int ring_dtype, int64_t ring_rows, int64_t row,
                    void *stream);

/* The same rows computed from the projections instead of observed: qk[h, r, f] = sum_d (q[r, h*hd+d] * s) *
 * (k[f, h*hd+d] * s), s = hd^-0.25, for n_sel heads and n_rows query rows, written to ring[slots[i]][row0 + r][:].
 * Lets the backend keep its fused attention (the reference must run every attention module unfused, inside
 * whisper.model.disable_sdpa(), just to read these rows: T.py:49-61, 903).
 *   q : device [n_rows][d_model]   (cross_attn.query output rows; dtype f32/f16)
 *   k : device [n_ctx][d_model]    (cross_attn.key output of the window)                                       */
WT_API int wt_qk_rows(const void *q, const void *k, int dtype, int n_rows, int n_ctx, int d_model, int head_dim, float scale,
               const int32_t *heads, const int32_t *slots, int n_sel, void *ring, int ring_dtype, int64_t ring_rows,
               int64_t row0, void *stream);

/* wt_qk_rows for a BATCH of independent 30 s windows and ALL hooked decoder layers in one launch: the batched
 * form of the naive strategy's teacher-forced re-run (T.py:1110-1121 hook, 1236-1249 forward + slice), where the
 * reference runs one window at a time.  For window b, selected head i (layer sel_layer[i], head sel_head[i]) and
 * query row r in [row_begin[b], row_end[b]):
 *     ring[b][sel_slot[i]][ring_row0 + r][f] = sum_d (q_l[b][r][h*hd+d] * s) * (k_l[b][f][h*hd+d] * s)
 *   q_layers_host / k_layers_host : HOST arrays of n_layers (<= WT_MAX_LAYERS) DEVICE pointers: cross_attn.query
 *                                   output (n_batch, n_q, d_model) and cross_attn.key output (n_batch, n_ctx, d_model)
 *                                   of each hooked layer; batch strides in elements; rows contiguous (d_model)
 *   sel_layer/sel_head/sel_slot   : device int32[n_sel]
 *   row_begin/row_end             : device int32[n_batch] or NULL (= all n_q rows)
 *   ring                          : device [n_batch][n_slots][ring_rows][n_ctx] (ring_batch_stride elements per window)
 * head_dim must be 64 (every Whisper checkpoint).  Arithmetic identical to wt_qk_rows. */
WT_API int wt_qk_rows_batch(const void *const *q_layers_host, const void *const *k_layers_host, int n_layers, int dtype, int n_batch,
                     int n_q, int64_t q_batch_stride, int64_t k_batch_stride, int n_ctx, int d_model, int head_dim,
                     float scale, const int32_t *sel_layer, const int32_t *sel_head, const int32_t *sel_slot, int n_sel,
                     const int32_t *row_begin, const int32_t *row_end, void *ring, int ring_dtype, int64_t ring_batch_stride,
                     int64_t ring_rows, int64_t ring_row0, void *stream);

/* wt_qk_rows_batch for B decoder STREAMS stepping together (the efficient strategy, T.py:783-793, with B independent
 * recordings in one decoder call instead of the reference's batch of one, T.py:806): every stream owns one block of a
 * (n_streams, n_slots, ring_rows, n_ctx) ring for its whole life, and the streams that take part in a given decoder
 * call are any subset of them: batch entry b writes ring block ring_index[b] (device int32[n_batch], required).
 * All n_q query rows are written at ring rows ring_row0 ... (the decode loop passes the last row only: n_q = 1 with
 * the layer pointers advanced to it).  Arithmetic identical to wt_qk_rows. */
WT_API int wt_qk_rows_streams(const void *const *q_layers_host, const void *const *k_layers_host, int n_layers, int dtype,
                              int n_batch, int n_q, int64_t q_batch_stride, int64_t k_batch_stride, int n_ctx, int d_model,
                              int head_dim, float scale, const int32_t *sel_layer, const int32_t *sel_head,
                              const int32_t *sel_slot, int n_sel, const int32_t *ring_index, void *ring, int ring_dtype,
                              int64_t ring_batch_stride, int64_t ring_rows, int64_t ring_row0, void *stream);

/* T.py:1540-1568.  For each unit: select heads, median filter (medfilt_width: 9 is the reference's default and what
 * every caller passes -- the tuned path; the other odd widths 1, 3, 5, 7 of the seam's parameter, T.py:1439, are
 * served by a per-class launch; even widths and widths > 9 are refused with WT_E_UNSUPPORTED;
 * scipy 'reflect' = half-sample symmetric edges) along frames, * qk_scale,
 * softmax over the F-frame window, mean over heads, divide by the per-frame L2
 * norm over tokens, negate, zero rows[:-1] of columns >= pad_from, then
 * cost[0,0] = min(cost).  fp32 arithmetic like the reference's torch CPU ops;
 * the result, widened to double, is the matrix the reference hands to dtw.dtw.
 *   qk        : QK logits (fp32 or fp16 per qk_dtype), layout per wt_seg_desc
 *   segs_host : n_seg descriptors in host memory (launch geometry).  Any order is correct; handing the units
 *               over sorted by ceil(F/256) lets the library launch each group over its own units only
 *   segs_dev  : the same n_seg descriptors in device memory
 *   head_idx  : device int32[n_heads], flat head indices in the order of
 *               alignment_heads.indices().T (T.py:1545); all heads => 0..L*H-1
 *   cost      : device fp32, written at segs[i].cost_offset (T*F each)       */
WT_API int wt_cost_batch(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                  const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, void *stream);

/* T.py:1572,1581 dtw.dtw(cost, step_pattern=symmetric1) + T.py:1648-1652.
 * dtw-python semantics (f64 accumulation; candidates diagonal, same-token/
 * previous-frame, previous-token/same-frame; strict '<', first wins; closed
 * ends), in-kernel backtrack.  Bit-exact integer outputs for a given cost.
 *   cost     : device fp32; MUST stay readable for 16 bytes past the end of the last unit (the sweep prefetches
 *              its cost rows with unconditional 16-byte loads; the extra floats are never used)
 *   jumps    : device int32, T+1 per unit at jumps_offset
 *   path_i/j : optional (may be NULL) device int32, warping path
 *              (alignment.index1s / index2s) at path_offset, forward order
 *   path_len : optional device int32[n_seg]
 *   dist     : optional device double[n_seg] = alignment.distance            */
WT_API int wt_dtw_batch(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg, int32_t *jumps,
                 int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, void *stream);

/* wt_dtw_batch with an explicit step pattern:
 *   WT_STEP_SYMMETRIC1        = dtw.stepPattern.symmetric1 (T.py:1572; what every caller of the reference uses)
 *   WT_STEP_NO_EMPTY_SUBWORDS = the pattern of T.py:1575-1580 (perform_word_alignment(subwords_can_be_empty=False)):
 *                               symmetric1 without the previous-token/same-frame move, so that two tokens never share
 *                               a timestamp; needs T <= F per unit (WT_E_UNSUPPORTED otherwise: dtw-python finds no path) */
#define WT_STEP_SYMMETRIC1 0
#define WT_STEP_NO_EMPTY_SUBWORDS 1
WT_API int wt_dtw_batch_pattern(const float *cost, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                         int step_pattern, int32_t *jumps, int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist,
                         void *stream);

/* wt_cost_batch followed by wt_dtw_batch on the same stream = perform_word_alignment's numerics (T.py:1540-1581,
 * 1648-1652) for a batch of units.  Units whose (T, F) matrix, direction planes and boundary rows fit a workgroup's 160 KB of LDS
 * (any T <= 256; e.g. T = 64 up to F = 480, T = 16 up to F = 1792: csrc/wt_small.h) -- the reference's
 * default per-segment call shape (T.py:544-557: T p50 11, F p50 144) -- leave the batched kernels after the row pass:
 * ONE fused kernel does their column norm, cost[0,0], DTW and backtrack in one workgroup with the matrix in LDS; which
 * path a unit takes depends on its own shape only, and cost / jumps are bit-identical on both.
 *   flags : WT_ALIGN_KEEP_COST             cost[] holds the final matrix of every unit (without it the small units
 *                                          leave only their head-mean rows there); wt_disfluency_batch needs it
 *           WT_ALIGN_NO_FUSED_SMALL_UNITS  every unit through the batched kernels (A/B measurements, tests)
 *           WT_ALIGN_ROWS_PER_CLASS        the row pass as one pipelined launch per F class even for a small batch
 *                                          (a small batch otherwise gets ONE launch that fetches eight heads at once;
 *                                          same bits either way: A/B measurements, tests)
 * wt_align_batch(...) = wt_align_batch_v3(..., WT_ALIGN_KEEP_COST, stream). */
#define WT_ALIGN_KEEP_COST 1
#define WT_ALIGN_NO_FUSED_SMALL_UNITS 2
#define WT_ALIGN_ROWS_PER_CLASS 4
WT_API int wt_align_batch_v3(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                      const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, int32_t *jumps,
                      int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, int flags, void *stream);
WT_API int wt_align_batch(const void *qk, int qk_dtype, const wt_seg_desc *segs_host, const wt_seg_desc *segs_dev, int n_seg,
                   const int32_t *head_idx, int n_heads, int medfilt_width, float qk_scale, float *cost, int32_t *jumps,
                   int32_t *path_i, int32_t *path_j, int32_t *path_len, double *dist, void *stream);

/* T.py:1656-1672 (detect_disfluencies): for every token row t of every unit, scipy.signal.find_peaks(-cost[t,
 * jumps[t]:jumps[t+1]], width=min_width, prominence=min_prominence) in f64 as scipy does; when more than one peak
 * survives, the token's start moves to round(left_ips[-1]) + jumps[t].
 *   cost, jumps : what wt_cost_batch / wt_dtw_batch wrote (same descriptors)
 *   jumps_start : device int32, T+1 per unit at jumps_offset: [t] = the (possibly moved) start of token t,
 *                 [T] = jumps[T].  The reference passes width=3, prominence=0.02. */
WT_API int wt_disfluency_batch(const float *cost, const wt_seg_desc *segs_dev, int n_seg, const int32_t *jumps,
                        int32_t *jumps_start, double min_prominence, double min_width, void *stream);

/* T.py:1795-1805 find_start_padding on a batch of (n_mels, n_cols) log-mel
 * windows: out[b] = None(-1) if the last column is not all-zero, else the
 * index after the last column in [1, n_cols-2] that differs from zero, else 0. */
WT_API int wt_find_start_padding_batch(const float *mel, int n_chunks, int n_mels, int n_cols, int32_t *out, void *stream);

/* Confidence path: T.py:871-876 (efficient: log_softmax of the filtered
 * logits, gather of the chosen token T.py:735) and T.py:1245,1292 (naive).
 * For each of n_rows rows of V fp32 logits (row r at logits + r*row_stride):
 *     out[r] = log_softmax(row with suppressed entries at -inf)[token[r]]
 *   (a NaN logit makes out[r] NaN, as F.log_softmax does; the reference asserts finiteness at T.py:736)
 *   suppress      : optional device uint8[n_rows_or_1][V] (1 = -inf), or NULL
 *   suppress_rows : 0 = none, 1 = one shared mask row, n_rows = per-row masks */
WT_API int wt_logprob_gather_batch(const void *logits, int logits_dtype, int64_t row_stride, int n_rows, int V,
                            const int32_t *token, const uint8_t *suppress, int suppress_rows, float *out, void *stream);

/* The same gather with an explicit row list: out[r] = log_softmax(logits row row_index[r])[token[r]], r < n_out.
 * Rows may repeat or be skipped: T.py:1292 `logprobs[:, step, tok]` for the text positions of many teacher-forced
 * windows at once, read from one padded (n_windows * T_max, V) logits block (no logit filters on this path, T.py:1245). */
WT_API int wt_logprob_gather_rows(const void *logits, int logits_dtype, int64_t row_stride, const int32_t *row_index, int n_out,
                           int V, const int32_t *token, float *out, void *stream);

/* The confidence path for B decoder STREAMS stepping together (T.py:849-881 with B recordings in one decoder call
 * instead of the reference's batch of one, T.py:806).  The reference keeps every step's filtered (V,) logits vector
 * (T.py:875-876) until the window closes; here, when a decoder call's rows are final (the sampler has filtered them in
 * place, the sampled tokens are known), ONE launch takes from each row everything the hook state machine can still
 * ask of it, and the row itself is not kept:
 *   digest[blk][ring_row][0]    log_softmax(row)[token]      (T.py:735; bit-identical to wt_logprob_gather_batch on a row at the same address modulo 128 bytes)
 *   digest[blk][ring_row][1..2] max(row), log(sum(exp(row - max))):  log_softmax(row)[t] = (row[t] - [1]) - [2]
 *   digest[blk][ring_row][3]    argmax(row) as int32 bits, first index of the maximum   (T.py:508,729,879)
 *   digest[blk][ring_row][4..7] row[aux_tokens[k]] (raw logits; -inf for k >= n_aux): <|endoftext|>, <|notimestamps|> ...
 *   slice[blk][ring_row][:]     row[slice_begin:V] (raw logits): the timestamp tokens -- T.py:535 takes
 *                               argmax(row[start_token + 1:]) with start_token a timestamp; optional (NULL = none)
 * with blk = ring_index[r] for batch row r.
 *   logits     : device fp32, row r at logits + r*row_stride (the (n_rows, n_q, V) decoder output's last position)
 *   token      : device int32 (token_dtype 0) or int64 (1), row r's token at token[r*token_stride]
 *   ring_index : device int32[n_rows]; digest: device fp32 [n_blocks][ring_rows][8]; slice: device fp32
 *                [n_blocks][ring_rows][V - slice_begin]
 *   aux_tokens_host : HOST int32[n_aux], n_aux <= 4 */
WT_API int wt_logprob_digest_streams(const float *logits, int64_t row_stride, int n_rows, int V, const void *token, int token_dtype,
                                     int64_t token_stride, const int32_t *ring_index, int64_t ring_rows, int64_t ring_row,
                                     const int32_t *aux_tokens_host, int n_aux, int slice_begin, float *digest, float *slice,
                                     void *stream);

/* openai-whisper audio.log_mel_spectrogram + pad_or_trim as called at
 * T.py:1213-1214 (naive path; n_frames = 3000) for a batch of equal-length
 * PCM chunks: hann-400 STFT (centre, reflect pad), hop 160, |.|^2, mel
 * (mel_fb: device fp32 [n_mels][201]), log10(max(.,1e-10)), max(x, max-8),
 * (x+4)/4; columns >= n_valid_frames[b] are exact zeros (pad_or_trim).
 *   mel_fb         : its banded form is cached per (stream, pointer, n_mels): do not modify the weights in place
 *                    while passing the same pointer (wt_shutdown() drops the cache)
 *   pcm            : device fp32 [n_chunks][n_samples]
 *   n_valid_samples: device int32[n_chunks] real samples per chunk (<= n_samples)
 *   mel_out        : device fp32 [n_chunks][n_mels][n_frames]
 *   gmax           : optional device fp32[n_chunks] out = per-chunk max of log10 mel (before the clamp)
 * n_frames may be anything (a whole file: n_samples / 160); the log10 uses the hardware log2 (|err| < 3e-7) and
 * the power is |X|^2 without the square root round trip of torch's abs()**2: results agree with
 * torch.stft-based log_mel_spectrogram to 2e-4 absolute (tests/test_gpu_parity.py). */
WT_API int wt_logmel_batch(const float *pcm, int n_chunks, int64_t n_samples, const int32_t *n_valid_samples, const float *mel_fb,
                    int n_mels, int n_frames, float *mel_out, float *gmax, void *stream);

/* wt_logmel_batch + find_start_padding (T.py:1795-1805) of the windows it has just written: start_of_padding[b]
 * (device int32[n_chunks]) = -1 (None: the last column is not all-zero), else the index after the last column in
 * [1, n_frames - 2] that is not all-zero, else 0 -- identical to wt_find_start_padding_batch(mel_out, ...) run
 * afterwards.  The columns behind the valid frames are zeros by construction of the finalising pass, so the
 * reference's backward walk starts at the last valid column instead of re-reading the zeros: one wave per window,
 * a few microseconds per batch instead of 18-65.  Everything it decides, it decides on the values in mel_out. */
WT_API int wt_logmel_pad_batch(const float *pcm, int n_chunks, int64_t n_samples, const int32_t *n_valid_samples,
                               const float *mel_fb, int n_mels, int n_frames, float *mel_out, float *gmax,
                               int32_t *start_of_padding, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WTALIGN_H */
