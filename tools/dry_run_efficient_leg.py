#!/usr/bin/env python3
"""bench.py's default-strategy leg (run_efficient_leg) on the CPU with the oracle-backed kernel stand-ins and the tiny
model: a check of the leg's own Python (scripts, parity bookkeeping, histogram keys) in the GPU-less container --
the numbers mean nothing.     python tools/dry_run_efficient_leg.py [streams]"""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from benchlib.default_strategy_leg import run_efficient_leg  # noqa: E402
import many_helper as H  # noqa: E402
import cpu_kernel_standin  # noqa: E402
from test_streams_host import install_streams_standin  # noqa: E402

patch = H._Patch()
cpu_kernel_standin.install(patch)
install_streams_standin(patch)
torch.cuda.synchronize = lambda *a, **k: None
args = types.SimpleNamespace(e2e_streams=int(sys.argv[1]) if len(sys.argv) > 1 else 4, no_cpu_baseline="--cpu-path" not in sys.argv,
                             e2e_device="cpu", e2e_islands=8, e2e_load_model=lambda device, attention="flat": H.load_tiny("cpu", attention),
                             e2e_cpu_parity_budget=60.0)
out = run_efficient_leg(args, lambda o: None)
print(json.dumps(out, indent=1)[:6000])
