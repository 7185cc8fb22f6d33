"""Pins the DTW leg of the oracle -- and, on a GPU box, the HIP kernel -- to the REAL dtw-python package.

dtw-python (requirements of the reference: /root/reference/requirements.txt:2; call sites transcribe.py:27,1572-1581,
1598,1648-1652) is absent from the build image and cannot be fetched (no network), so `oracle/dtw_ref.c` restates its
published algorithm and every golden's DTW leg goes through that restatement ("parity unpinned").
This file is the one command that turns "unpinned" into "pinned" on any machine that has the wheel:

    pip install dtw-python && python -m pytest tests/test_pin_dtw_python.py -q            # the oracle (CPU)
    pip install dtw-python && python -m pytest tests/test_pin_dtw_python.py -q -m gpu     # + libwtalign.so on an MI355X

It skips (and says so) where the package is missing.  What is compared: index1s, index2s (the warping path the
reference reads), the jumps the reference derives from them (transcribe.py:1648-1652), and the distance -- for the
reference's two step patterns (symmetric1, transcribe.py:1572; the custom pattern of transcribe.py:1575-1580) on the
inputs where the TIE ORDER decides the path: all-zero matrices, quantised costs with exact ties, and the zero plateau
the reference's pad mask `weights[:-1, max_duration:] = 0` (transcribe.py:1565) creates on the last window of every
file, at the full (224, 1500) window size.
"""
import numpy as np
import pytest

dtw = pytest.importorskip("dtw", reason="dtw-python is not installed: `pip install dtw-python` pins the DTW oracle (and, "
                                        "with -m gpu, the HIP kernel) to the package the reference calls")

from oracle import align_ref as O  # noqa: E402


def real_dtw(cost, pattern):
    """Exactly the reference's call (transcribe.py:1571-1581)."""
    if pattern == 0:
        step_pattern = dtw.stepPattern.symmetric1
    else:
        step_pattern = dtw.stepPattern.StepPattern(dtw.stepPattern._c(
            1, 1, 1, -1,
            1, 0, 0, 1,
            2, 0, 1, -1,
            2, 0, 0, 1,
        ))
    return dtw.dtw(np.asarray(cost, dtype=np.float64), step_pattern=step_pattern)


def reference_jumps(alignment):
    """transcribe.py:1648-1652, verbatim in effect."""
    jumps = np.diff(alignment.index1s)
    jumps = np.pad(jumps, (1, 0), constant_values=1)
    jumps = jumps.astype(bool)
    jumps = alignment.index2s[jumps]
    return np.pad(jumps, (0, 1), constant_values=alignment.index2s[-1])


def product_like_cost(rng, T, F, pad_from=None, quantise=None):
    """A matrix with the product's invariants: all <= 0, column-normalised look, cost[0,0] = min; optionally the pad
    mask's zero plateau (rows[:-1], columns >= pad_from) and values snapped to a grid (exact ties)."""
    c = -rng.rand(T, F)
    ridge = np.sort(rng.randint(0, F, size=T))
    for t in range(T):
        c[t, max(ridge[t] - 1, 0):ridge[t] + 2] -= 1.0
    if quantise:
        c = np.round(c * quantise) / quantise
    if pad_from is not None:
        c[:-1, pad_from:] = 0.0
    c[0, 0] = c.min()
    return c


def corpus():
    rng = np.random.RandomState(20240930)
    out = []
    # all-equal costs: every candidate ties at every cell
    for T, F in ((1, 1), (1, 7), (3, 5), (5, 5), (7, 64), (64, 65), (224, 1500)):
        out.append((f"zeros_{T}x{F}", np.zeros((T, F))))
    out.append(("constant_negative_9x40", np.full((9, 40), -0.25)))
    # exact ties from quantisation
    for k in range(12):
        T, F = int(rng.randint(2, 40)), int(rng.randint(40, 300))
        out.append((f"quantised_{k}_{T}x{F}", product_like_cost(rng, T, F, quantise=[2, 4, 8, 16][k % 4])))
    # the pad-mask plateau (transcribe.py:1565) at realistic and at full size
    for k in range(6):
        T, F = int(rng.randint(5, 60)), int(rng.randint(100, 600))
        out.append((f"plateau_{k}_{T}x{F}", product_like_cost(rng, T, F, pad_from=int(rng.randint(F // 2, F)))))
    out.append(("plateau_full_window_224x1500", product_like_cost(rng, 224, 1500, pad_from=811)))
    out.append(("plateau_full_window_quantised_224x1500", product_like_cost(rng, 224, 1500, pad_from=1203, quantise=8)))
    # generic (tie-free) inputs incl. mixed sign, for completeness
    for k in range(6):
        T, F = int(rng.randint(1, 50)), int(rng.randint(50, 400))
        out.append((f"generic_{k}_{T}x{F}", rng.standard_normal((T, F))))
    return out


CORPUS = corpus()


@pytest.mark.parametrize("pattern", [0, 1], ids=["symmetric1", "no_empty_subwords"])
@pytest.mark.parametrize("name,cost", CORPUS, ids=[n for n, _ in CORPUS])
def test_oracle_dtw_equals_dtw_python(name, cost, pattern):
    T, F = cost.shape
    if pattern == 1 and T > F:
        pytest.skip("the second pattern has no path when T > F")
    want = real_dtw(cost, pattern)
    got = O.dtw_ref(cost, step_pattern=pattern)
    assert np.array_equal(got.index1s, np.asarray(want.index1s)), name
    assert np.array_equal(got.index2s, np.asarray(want.index2s)), name
    assert got.distance == float(want.distance), name
    assert np.array_equal(O.jumps_from_path(got.index1s, got.index2s), reference_jumps(want)), name


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", [0, 1], ids=["symmetric1", "no_empty_subwords"])
def test_hip_dtw_equals_dtw_python(pattern):
    """wt_dtw_batch_pattern on the whole corpus in ONE launch vs the package, unit by unit: path, jumps, distance."""
    import torch
    from whisper_timestamped import _lib
    items = [(n, c) for n, c in CORPUS if not (pattern == 1 and c.shape[0] > c.shape[1])]
    dev = torch.device("cuda", 0)
    descs = _lib.make_descs(len(items))
    for d, (_, c) in zip(descs, items):
        d["T"], d["F"], d["start_token"], d["pad_from"] = c.shape[0], c.shape[1], 0, -1
    n_cost, n_jumps, n_path = _lib.layout_outputs(descs)
    cost = torch.zeros(n_cost, dtype=torch.float32)
    for d, (_, c) in zip(descs, items):
        # the kernel's input is fp32 (the reference's matrix IS fp32 values widened to double, transcribe.py:1550)
        c0 = int(d["cost_offset"])
        cost[c0:c0 + c.size] = torch.from_numpy(c.astype(np.float32).reshape(-1))
    cost = cost.to(dev)
    jumps = torch.empty(n_jumps, dtype=torch.int32, device=dev)
    pi = torch.empty(n_path, dtype=torch.int32, device=dev)
    pj = torch.empty(n_path, dtype=torch.int32, device=dev)
    plen = torch.empty(len(items), dtype=torch.int32, device=dev)
    dist = torch.empty(len(items), dtype=torch.float64, device=dev)
    _lib.dtw_batch(cost, descs, _lib.descs_to_device(descs, dev), jumps, pi, pj, plen, dist, step_pattern=pattern)
    torch.cuda.synchronize()
    jumps, pi, pj, plen, dist = jumps.cpu().numpy(), pi.cpu().numpy(), pj.cpu().numpy(), plen.cpu().numpy(), dist.cpu().numpy()
    for k, (d, (name, c)) in enumerate(zip(descs, items)):
        want = real_dtw(c.astype(np.float32), pattern)
        p0, n = int(d["path_offset"]), int(plen[k])
        assert np.array_equal(pi[p0:p0 + n], np.asarray(want.index1s)), name
        assert np.array_equal(pj[p0:p0 + n], np.asarray(want.index2s)), name
        j0 = int(d["jumps_offset"])
        assert np.array_equal(jumps[j0:j0 + c.shape[0] + 1], reference_jumps(want)), name
        assert dist[k] == float(want.distance), name
