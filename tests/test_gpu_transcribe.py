"""transcribe() end to end on the MI355X against the REFERENCE'S OWN OUTPUT.

tests/golden/transcribe_cases.json holds what /root/reference's unmodified
transcribe_timestamped produced on the CPU for each case (same whisper double,
same random-init model, same audio, same sampled tokens).  Here this
repository's transcribe() runs on the GPU: model forward by torch (hipBLASLt),
attention capture / cost / DTW / log-prob gather / log-mel by libwtalign.so.

Bars (BASELINE.json north_star): word start/end within +-0.02 s, confidences
within 1e-4 before the reference's round(,3) -- compared after rounding here, so
a rounding flip may show as 1e-3; texts, tokens and segmentation identical.
"""
import copy
import json
import os

import pytest
import torch

from golden import make_golden_transcribe as G
from test_transcribe_host import CASES, compare, run_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_transcribe_matches_reference_output(case):
    got = run_case(copy.deepcopy(case), device="cuda:0")
    dt, dc = compare(got, case["expected"], time_tol=0.02, conf_tol=1e-3 + 1e-4, logprob_tol=2e-4)
    print(f"{case['name']}: max |dt| = {dt:.3f} s, max |dconfidence| = {dc:.4f}")


def test_capture_ring_rows_match_reference_hook():
    """wt_capture_rows == what hook_attention_weights keeps (transcribe.py:783-793), for the selected heads."""
    from whisper_timestamped.capture import QKCaptureRing
    g = torch.Generator().manual_seed(3)
    pairs = [(0, 1), (1, 0), (1, 5), (3, 2)]
    ring = QKCaptureRing("cuda:0", pairs, n_hooked_layers=4, n_heads=6, n_ctx=1500, capacity=16)
    ring16 = QKCaptureRing("cuda:0", pairs, n_hooked_layers=4, n_heads=6, n_ctx=1500, capacity=16, dtype=torch.float16)
    kept = {}
    for row, n_q in enumerate([5, 1, 1, 3]):
        for layer in range(4):
            qk = torch.randn((1, 6, n_q, 1500), generator=g)
            ring.write(layer, qk.cuda(), row)
            ring16.write(layer, qk.cuda(), row)
            kept[(layer, row)] = qk[:, :, -1:, :]
    torch.cuda.synchronize()
    for slot, (l, h) in enumerate(pairs):
        for row in range(4):
            assert torch.equal(ring.buf[slot, row].cpu(), kept[(l, row)][0, h, 0])
            assert torch.equal(ring16.buf[slot, row].cpu(), kept[(l, row)][0, h, 0].half())
    view = ring.rows([1, 2, 3])
    assert view.data_ptr() == ring.buf[:, 1].data_ptr() and view.shape == (4, 3, 1500)
    gathered = ring.rows([0, 2])
    assert torch.equal(gathered[:, 1], ring.buf[:, 2])
