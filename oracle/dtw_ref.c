/*
 * oracle/dtw_ref.c -- TEST INFRASTRUCTURE ONLY (the CPU oracle).
 *
 * Plain-C restatement of the DTW the reference calls at
 *   whisper_timestamped/transcribe.py:1572,1581
 *       alignment = dtw.dtw(weights, step_pattern=dtw.stepPattern.symmetric1)
 * and consumes at transcribe.py:1598,1648-1652 (alignment.index1s / index2s).
 *
 * The arithmetic lives in the third-party package dtw-python (UNPINNED in the
 * reference: requirements.txt:2, setup.py:7), which is absent from
 * /root/reference and from this image.  What is restated here is its published
 * algorithm (dtw/dtw_core.c:computeCM + dtw/_backtrack.py + stepPattern
 * symmetric1 = _c(1,1,1,-1, 1,0,0,1, 2,0,1,-1, 2,0,0,1, 3,1,0,-1, 3,0,0,1)):
 *
 *   - local cost lm is given directly (y=None), rows = tokens (query, index1),
 *     columns = frames (reference series, index2), double precision;
 *   - cm = NaN everywhere, cm[0,0] = lm[0,0]; sweep "for j in frames: for i in
 *     tokens"; a cell that is already non-NaN is skipped;
 *   - three candidate sums, in pattern order
 *        p1 = cm[i-1,j-1] + 1.0*lm[i,j]    (diagonal)
 *        p2 = cm[i  ,j-1] + 1.0*lm[i,j]    (same token, previous frame)
 *        p3 = cm[i-1,j  ] + 1.0*lm[i,j]    (previous token, same frame)
 *     out-of-range predecessors leave the candidate NaN;
 *   - argmin starts from +INFINITY and uses strict '<' : first minimum wins,
 *     NaN never wins, the SUMS are compared (not the predecessors);
 *   - sm[i,j] = winning pattern number (1..3);
 *   - backtrack from (T-1,F-1) following sm to (0,0), prepending.
 *
 * PARITY STATUS: "parity unpinned" against dtw-python itself (it cannot be run
 * here); pinned instead against (a) exhaustive minimum-cost path enumeration
 * on small matrices, (b) transformers' independent _dynamic_time_warping on
 * tie-free inputs, and (c) the reference's own call sites driven through this
 * code (tests/golden/make_golden.py).  See DESIGN.md section "Oracle".
 *
 * Nothing in the shipped product may link or call this file: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LM(i, j) lm[(size_t)(i) * (size_t)F + (size_t)(j)]
#define CM(i, j) cm[(size_t)(i) * (size_t)F + (size_t)(j)]
#define SM(i, j) sm[(size_t)(i) * (size_t)F + (size_t)(j)]

static int argmin3(const double *c) {
    /* dtw_core.c:argmin -- strict '<' against a running +inf */
    int best = -1;
    double v = INFINITY;
    for (int k = 0; k < 3; ++k) {
        if (c[k] < v) {
            best = k;
            v = c[k];
        }
    }
    return best;
}

/* Global cost + direction matrices, row-major (T,F).  step_pattern: 0 =
 * symmetric1 (transcribe.py:1572); 1 = the custom pattern of
 * transcribe.py:1575-1580 (no previous-token/same-frame move). */
int wt_ref_dtw_cm(const double *lm, int T, int F, int step_pattern, double *cm, int32_t *sm) {
    if (T <= 0 || F <= 0) return -1;
    for (size_t k = 0; k < (size_t)T * (size_t)F; ++k) {
        if (isnan(lm[k])) return -4; /* dtw-python rejects NaN local costs */
        cm[k] = NAN;
        sm[k] = INT32_MIN; /* R's NA_integer_, what dtw-python stores for "no step" */
    }
    CM(0, 0) = LM(0, 0);
    const int npat = step_pattern == 0 ? 3 : 2;
    for (int j = 0; j < F; ++j) {
        for (int i = 0; i < T; ++i) {
            if (!isnan(CM(i, j))) continue;
            double cl[3] = {NAN, NAN, NAN};
            if (i >= 1 && j >= 1) { cl[0] = CM(i - 1, j - 1); cl[0] += 1.0 * LM(i, j); }
            if (j >= 1)           { cl[1] = CM(i, j - 1);     cl[1] += 1.0 * LM(i, j); }
            if (npat == 3 && i >= 1) { cl[2] = CM(i - 1, j); cl[2] += 1.0 * LM(i, j); }
            int m = argmin3(cl);
            if (m > -1) {
                CM(i, j) = cl[m];
                SM(i, j) = m + 1;
            }
        }
    }
    return 0;
}

/* dtw/_backtrack.py restated; writes the path in forward order.
 * idx1/idx2 must hold T+F-1 entries.  Returns the path length or <0. */
int wt_ref_backtrack(const int32_t *sm, int T, int F, int32_t *idx1, int32_t *idx2) {
    int cap = T + F - 1;
    int n = 0;
    int i = T - 1, j = F - 1;
    /* fill from the back, then shift */
    int pos = cap;
    idx1[--pos] = i;
    idx2[pos] = j;
    n = 1;
    while (!(i == 0 && j == 0)) {
        int32_t s = SM(i, j);
        if (s == INT32_MIN) return -5; /* no path (cannot happen for symmetric1 on finite costs) */
        if (s == 1) { i -= 1; j -= 1; }
        else if (s == 2) { j -= 1; }
        else if (s == 3) { i -= 1; }
        else return -6;
        if (i < 0 || j < 0 || pos == 0) return -7;
        idx1[--pos] = i;
        idx2[pos] = j;
        ++n;
    }
    memmove(idx1, idx1 + pos, (size_t)n * sizeof(int32_t));
    memmove(idx2, idx2 + pos, (size_t)n * sizeof(int32_t));
    return n;
}

/* transcribe.py:1648-1652:
 *   jumps = np.diff(index1s); pad (1,0) with 1; astype(bool);
 *   jumps = index2s[jumps]; pad (0,1) with index2s[-1]
 * jumps_out must hold T+1 entries; returns the count written. */
int wt_ref_jumps(const int32_t *idx1, const int32_t *idx2, int n, int32_t *jumps_out) {
    int k = 0;
    for (int p = 0; p < n; ++p) {
        int d = (p == 0) ? 1 : (idx1[p] - idx1[p - 1]);
        if (d != 0) jumps_out[k++] = idx2[p];
    }
    jumps_out[k++] = idx2[n - 1];
    return k;
}

/* Convenience: cost(T,F) f64 -> jumps(T+1), path, distance.  Scratch is
 * allocated here.  path_i/path_j/path_len/dist may be NULL. */
int wt_ref_dtw_jumps(const double *lm, int T, int F, int32_t *jumps_out, int32_t *path_i, int32_t *path_j,
                     int32_t *path_len, double *dist) {
    double *cm = (double *)malloc((size_t)T * F * sizeof(double));
    int32_t *sm = (int32_t *)malloc((size_t)T * F * sizeof(int32_t));
    int32_t *i1 = (int32_t *)malloc((size_t)(T + F) * sizeof(int32_t));
    int32_t *i2 = (int32_t *)malloc((size_t)(T + F) * sizeof(int32_t));
    int rc = -8;
    if (cm && sm && i1 && i2) {
        rc = wt_ref_dtw_cm(lm, T, F, 0, cm, sm);
        if (rc == 0) {
            int n = wt_ref_backtrack(sm, T, F, i1, i2);
            if (n < 0) rc = n;
            else {
                int k = wt_ref_jumps(i1, i2, n, jumps_out);
                rc = (k == T + 1) ? 0 : -9;
                if (path_i) memcpy(path_i, i1, (size_t)n * sizeof(int32_t));
                if (path_j) memcpy(path_j, i2, (size_t)n * sizeof(int32_t));
                if (path_len) *path_len = n;
                if (dist) *dist = CM(T - 1, F - 1);
            }
        }
    }
    free(cm); free(sm); free(i1); free(i2);
    return rc;
}

/* Exhaustive minimum path cost over monotone paths with moves (1,1),(0,1),(1,0)
 * via plain recursion with memo -- an independent check of cm[T-1,F-1] used by
 * tests on tiny matrices (same recurrence, different evaluation order:
 * row-major, min of predecessors first).  Returns the optimum cost. */
double wt_ref_dtw_bruteforce_cost(const double *lm, int T, int F) {
    double *g = (double *)malloc((size_t)T * F * sizeof(double));
    for (int i = 0; i < T; ++i)
        for (int j = 0; j < F; ++j) {
            double best = INFINITY;
            if (i == 0 && j == 0) best = 0.0;
            if (i > 0 && j > 0 && g[(size_t)(i - 1) * F + j - 1] < best) best = g[(size_t)(i - 1) * F + j - 1];
            if (j > 0 && g[(size_t)i * F + j - 1] < best) best = g[(size_t)i * F + j - 1];
            if (i > 0 && g[(size_t)(i - 1) * F + j] < best) best = g[(size_t)(i - 1) * F + j];
            g[(size_t)i * F + j] = best + LM(i, j);
        }
    double r = g[(size_t)T * F - 1];
    free(g);
    return r;
}
