"""TEST INFRASTRUCTURE (oracle): a second, structurally different restatement of dtw-python's DTW -- a generic
step-pattern INTERPRETER in pure Python.  Parity status: dtw-python itself is absent from the image (unpinned in
/root/reference/requirements.txt:2), so this is pinned to the package's published algorithm, not to its execution
("parity unpinned" for the DTW core).  What it adds over ``oracle/dtw_ref.c``: that file hard-codes
the two patterns as three (two) additions in a fixed order; this one CONSUMES the pattern rows exactly as the reference
hands them to the package -- ``dtw.stepPattern._c(1,1,1,-1, 1,0,0,1, 2,0,1,-1, 2,0,0,1)`` at
/root/reference/whisper_timestamped/transcribe.py:1575-1580, and the published rows of ``symmetric1`` -- and runs
dtw-python's generic loops over them:

  computeCM (dtw/dtw_core.c):  for j in frames: for i in tokens: every row (pattern p, di, dj, w) with (i-di, j-dj) inside
      the matrix contributes to candidate p -- weight -1: the candidate STARTS from the accumulated cost cm[i-di, j-dj];
      any other weight: += w * lm[i-di, j-dj]; candidates of patterns whose start fell outside stay NaN; the cell takes the
      argmin with strict `<` in pattern order (first wins on ties, NaN never wins), direction = pattern number.
  backtrack (dtw/_backtrack.py): from (T-1, F-1), follow the direction's start row back to (0, 0), prepending.

Only tests/, tests/golden/make_golden*.py and bench.py's checker code may import this module.
"""
import math

import numpy as np

# dtw-python's published `symmetric1` (dtw/stepPattern.py): diagonal, same token / previous frame, previous token / same frame
SYMMETRIC1_ROWS = (1, 1, 1, -1,
                   1, 0, 0, 1,
                   2, 0, 1, -1,
                   2, 0, 0, 1,
                   3, 1, 0, -1,
                   3, 0, 0, 1)


class StepPattern:
    """What ``dtw.stepPattern.StepPattern(dtw.stepPattern._c(...))`` carries: rows of (pattern, di, dj, weight)."""

    def __init__(self, rows):
        flat = [float(x) for x in rows]
        assert len(flat) % 4 == 0 and flat, "pattern rows are (pattern number, token step, frame step, weight)"
        self.rows = [(int(flat[k]), int(flat[k + 1]), int(flat[k + 2]), flat[k + 3]) for k in range(0, len(flat), 4)]
        self.n_patterns = max(r[0] for r in self.rows)
        for p in range(1, self.n_patterns + 1):
            mine = [r for r in self.rows if r[0] == p]
            assert mine and mine[0][3] == -1.0 and all(r[3] != -1.0 for r in mine[1:]), f"pattern {p}: one start row (weight -1), first"

    def start_of(self, p):
        return next(r for r in self.rows if r[0] == p)


def _c(*rows):
    """``dtw.stepPattern._c``: the flat argument list, kept as given (StepPattern reshapes it)."""
    return tuple(rows)


symmetric1 = StepPattern(SYMMETRIC1_ROWS)


def compute_cm(lm, pattern: StepPattern):
    """-> (cm, sm): accumulated cost (NaN = unreachable) and direction matrix (0 = none), both (T, F) Python lists."""
    T, F = len(lm), len(lm[0])
    nan = float("nan")
    cm = [[nan] * F for _ in range(T)]
    sm = [[0] * F for _ in range(T)]
    cm[0][0] = lm[0][0]
    rows, npat = pattern.rows, pattern.n_patterns
    for j in range(F):
        for i in range(T):
            if not math.isnan(cm[i][j]):           # (0, 0) is already set
                continue
            clist = [nan] * npat
            for p, di, dj, w in rows:
                ii, jj = i - di, j - dj
                if ii >= 0 and jj >= 0:
                    if w == -1.0:
                        clist[p - 1] = cm[ii][jj]
                    else:
                        clist[p - 1] += w * lm[ii][jj]
            best, val = -1, math.inf
            for k, c in enumerate(clist):          # strict `<`: the first minimum wins, NaN never does
                if c < val:
                    best, val = k, c
            if best >= 0:
                cm[i][j] = val
                sm[i][j] = best + 1
    return cm, sm


def backtrack(sm, pattern: StepPattern):
    i, j = len(sm) - 1, len(sm[0]) - 1
    i1, i2 = [i], [j]
    while not (i == 0 and j == 0):
        s = sm[i][j]
        if s == 0:
            raise ValueError("No warping path found compatible with the local constraints")
        _, di, dj, _ = pattern.start_of(s)
        i, j = i - di, j - dj
        i1.insert(0, i)
        i2.insert(0, j)
    return i1, i2


class Alignment:
    """The attributes of dtw-python's result object the reference reads (transcribe.py:1598,1648-1652)."""

    def __init__(self, index1s, index2s, distance):
        self.index1s = np.asarray(index1s, dtype=np.int32)
        self.index2s = np.asarray(index2s, dtype=np.int32)
        self.index1, self.index2 = self.index1s, self.index2s      # single-step patterns
        self.distance = distance


def dtw(x, y=None, step_pattern=symmetric1, **kw):
    """``dtw.dtw(local_cost_matrix, step_pattern=...)`` for the call shape of transcribe.py:1581 (y=None: x IS the local
    cost, rows = tokens, columns = frames; closed begin and end, no window)."""
    assert y is None and not kw.get("open_end") and not kw.get("open_begin")
    if isinstance(step_pattern, (tuple, list)):
        step_pattern = StepPattern(step_pattern)
    lm = np.asarray(x, dtype=np.float64)
    if lm.ndim != 2:
        raise ValueError("local cost must be 2-D")
    if np.isnan(lm).any():
        raise ValueError("NaN in local cost matrix")
    cm, sm = compute_cm(lm.tolist(), step_pattern)
    i1, i2 = backtrack(sm, step_pattern)
    return Alignment(i1, i2, cm[-1][-1])


def stub_modules(on_call=None, cross_check=None):
    """``dtw`` / ``dtw.stepPattern`` module objects backed by this interpreter, for loading the reference's UNMODIFIED
    transcribe.py where dtw-python is absent (tests/golden/make_golden*.py): the reference's own
    ``dtw.stepPattern.StepPattern(dtw.stepPattern._c(...))`` expression is EXECUTED against these, not recognised.
    on_call(local_cost, result): observer (the generators record the cost matrix the reference built).
    cross_check(local_cost, pattern) -> (index1s, index2s): a second implementation every call is held against."""
    import types
    d, sp = types.ModuleType("dtw"), types.ModuleType("dtw.stepPattern")
    sp.symmetric1, sp._c, sp.StepPattern = symmetric1, _c, StepPattern

    def call(x, y=None, step_pattern=symmetric1, **kw):
        res = dtw(x, y, step_pattern=step_pattern, **kw)
        if cross_check is not None:
            i1, i2 = cross_check(x, step_pattern)
            assert np.array_equal(i1, res.index1s) and np.array_equal(i2, res.index2s), "the two DTW restatements disagree"
        if on_call is not None:
            on_call(x, res)
        return res
    d.dtw, d.stepPattern = call, sp
    return {"dtw": d, "dtw.stepPattern": sp}
