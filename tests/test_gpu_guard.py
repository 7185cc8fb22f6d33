"""Every kernel of libwtalign.so on buffers fenced by unmapped pages (tests/guard/): an access one element outside
any input or output is a GPU memory access fault.  Each case runs in a child process -- a fault kills the child, not
the test session -- and must produce results bit-identical to the same call on ordinary tensors."""
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "guard", "libwtguard.so")),
                                 reason="tests/guard/libwtguard.so was not built (__graft_entry__.build() reports why)")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNNER = os.path.join(ROOT, "tests", "guard", "run_guarded.py")
CASES = ["step_kfull", "step_kreal", "step_largev3_fp16", "odd_units_f32", "odd_units_f16", "odd_units_f32_batched_only", "odd_units_f16_batched_only", "odd_units_f32_rows_per_class", "odd_units_f16_rows_per_class", "logprob", "digest", "logmel", "capture"]


@pytest.mark.parametrize("mode", ["end", "start"])
@pytest.mark.parametrize("case", CASES)
def test_fenced_buffers(case, mode):
    p = subprocess.run([sys.executable, RUNNER, case, mode], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, f"{case}/{mode}: exit {p.returncode}\n{p.stdout[-1500:]}\n{p.stderr[-3000:]}"
    assert f"guarded {case}/{mode}: ok" in p.stdout
