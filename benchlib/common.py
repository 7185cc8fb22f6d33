"""What every leg shares: paths, the partial-result emitter, JSON helpers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured float4 copy; tools/probes/read_probe: 6.6 read-only)
METRIC = "audio-seconds aligned/sec (whole node), whisper-base 30s chunks"


def json_scalar(o):
    """numpy scalars that slipped into a result dictionary"""
    if hasattr(o, "item"):
        return o.item()
    raise TypeError(f"Object of type {o.__class__.__name__} is not JSON serializable")


def make_emitter(path):
    """Children publish their (partial) results by atomically rewriting one JSON file: whatever was measured before
    a GPU fault is still there for the parent."""
    def emit(obj):
        if not path:
            return
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(obj, f, default=json_scalar)
        os.replace(tmp, path)
    return emit


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)
