"""Picklable helpers for sharding.transcribe_many's worker processes (test / bench infrastructure): the whisper double
as the model, and a scripted ~110-token transcript per 30 s clip (the same script tools/bench_transcribe.py times)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def load_base(device):
    import whisper_double as W
    W.install()
    return W.build_model("base", seed=0, device=device)


def load_tiny(device):
    import whisper_double as W
    W.install()
    return W.build_model("tiny", seed=0, device=device)


SEGMENTS = [(0, 20, 280), (300, 22, 600), (620, 18, 900), (920, 21, 1200), (1220, 19, 1490)]   # (start frame, text tokens, end frame)


def script_clip(index):
    """Before item `index`: the decoder of the double follows a scripted transcript (5 timestamped segments, ~110 tokens)."""
    from whisper_double.decoding import Script, set_script
    from golden import make_golden_transcribe as G
    segs = [(s, [None] * n, e) for s, n, e in SEGMENTS]
    set_script(Script([G.window_script(50364, 50257, segs, "eot")]))


def script_batch(indices):
    """transcribe_many(streams=N): the scripted transcript for every recording of the worker's batch (one window each)."""
    from whisper_double.decoding import Script, set_row_scripts
    from golden import make_golden_transcribe as G
    from whisper_timestamped import streams
    window = G.window_script(50364, 50257, [(s, [None] * n, e) for s, n, e in SEGMENTS], "eot")
    scripts = [Script([window]) for _ in indices]

    def on_group(rows):
        for r in rows:
            scripts[r].begin_window()
        set_row_scripts([scripts[r] for r in rows])
    streams.ON_GROUP_DECODE = on_group


class _Patch:
    """the two methods of pytest's monkeypatch that cpu_kernel_standin.install uses (a worker process has no fixture)"""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


class _Undo(_Patch):
    """...and its undo (bench.py's CPU baseline of the default strategy installs the stand-in for one leg only)."""

    def __init__(self):
        self.saved = []

    def setattr(self, obj, name, value):
        self.saved.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    def undo(self):
        for obj, name, value in reversed(self.saved):
            setattr(obj, name, value)
        self.saved = []


_STANDIN = False


def script_clip_cpu(index):
    """CPU host-logic tests: the oracle-backed stand-in for the kernels (tests/cpu_kernel_standin.py), installed once per
    worker process, then the scripted transcript."""
    global _STANDIN
    if not _STANDIN:
        import torch
        import cpu_kernel_standin
        torch.set_num_threads(4)          # (several workers share the host's cores)
        cpu_kernel_standin.install(_Patch())
        _STANDIN = True
    script_clip(index)
