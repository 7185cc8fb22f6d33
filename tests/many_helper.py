"""Picklable helpers for sharding.transcribe_many's worker processes (test / bench infrastructure): the whisper double
as the model, and a scripted ~110-token transcript per 30 s clip (the same script tools/bench_transcribe.py times)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "whisper-timestamped_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def load_base(device, attention="flat"):
    import whisper_double as W
    W.install()
    return W.build_model("base", seed=0, device=device, attention=attention)


def load_base_peaked(device):
    """whisper-base dims whose alignment heads attend along a monotone ridge (whisper_double.model.sharpen_cross_attention)."""
    return load_base(device, attention="peaked")


def load_tiny(device, attention="flat"):
    import whisper_double as W
    W.install()
    return W.build_model("tiny", seed=0, device=device, attention=attention)


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


# ---- ragged scripted transcripts (bench.py's ragged legs; also built inside worker processes, hence here)
def ragged_window(rs, frames, ts0, eot, lo=40, hi=160):
    """One window's scripted transcript with its OWN shape: 2-9 timestamped segments, 40-160 text tokens in all (scaled
    down for a short window), segment lengths and pauses drawn at random inside `frames` 20 ms frames.  Text tokens: two
    scripted word pieces, then `None` = "the model's most likely text token" (log-probabilities that mean something); the
    first timestamp respects max_initial_timestamp (1 s), timestamps never decrease (the sampler's rules: a scripted token
    the filters suppress would have log-probability -inf)."""
    import numpy as np
    from golden import make_golden_transcribe as G
    n_seg = int(rs.randint(2, 10))
    total = int(rs.randint(lo, hi + 1) * min(1.0, frames / 1500.0 + 0.2))
    total = max(total, n_seg)
    share = rs.dirichlet(np.full(n_seg, 2.0))
    usable = max(frames - 60, 10 * n_seg)
    segs, t = [], int(rs.randint(0, min(50, max(1, frames // 10))))
    for k in range(n_seg):
        dur = max(6, int(share[k] * usable))
        n_tok = max(3, int(round(share[k] * total)))
        s0, e0 = t, min(t + dur, frames - 1)
        if e0 - s0 < 4:
            break
        # (two scripted word pieces first: a segment whose most-likely tokens all happen to be punctuation has no words, and
        #  the reference's state machine -- T.py:1002-1018 `reset(add_segment=False)` -- then loses the next segment's start)
        segs.append((s0, G.text_ids(int(rs.randint(1 << 30)), 2) + [None] * (n_tok - 2), e0))
        t = min(e0 + int(rs.randint(0, 12)), frames - 2)
    return G.window_script(ts0, eot, segs, "eot")


def peaked_segments(counts, first_pos=3, stride=None, pause=0):
    """[(start frame, n text tokens, end frame)] for segments of `counts` text tokens whose timestamps follow the ridge of
    the "peaked" double: the script token at decoder position q is predicted from position q - 1, whose cross-attention
    peaks at frame (q - 1) * stride; a segment <|s|> text x n <|e|> starting at position q0 spans frames
    [q0 * stride, (q0 + n + 1) * stride].  first_pos = length of the prompt in front of the script (3: sot, language, task);
    pause = extra positions' worth of silence between segments (moves the timestamps, not the ridge)."""
    from whisper_double.model import PEAK_STRIDE
    stride = stride or PEAK_STRIDE
    segs, q = [], first_pos
    for n in counts:
        segs.append((q * stride, int(n), (q + int(n) + 1) * stride))
        q += int(n) + 2
    return segs


def peaked_window(rs, frames, ts0, eot, lo=40, hi=160, first_pos=3):
    """ragged_window for the "peaked" double: 2-9 segments, 40-160 text tokens (fewer where the window is short: one
    position is PEAK_STRIDE frames), timestamps placed where the model's attention ridge is (peaked_segments)."""
    import numpy as np
    from golden import make_golden_transcribe as G
    from whisper_double.model import PEAK_STRIDE
    n_seg = int(rs.randint(2, 10))
    room = (frames - 2) // PEAK_STRIDE - first_pos - 2 * n_seg - 1          # positions left for text tokens
    while room < 3 * n_seg and n_seg > 1:
        n_seg -= 1
        room = (frames - 2) // PEAK_STRIDE - first_pos - 2 * n_seg - 1
    total = max(3 * n_seg, min(int(rs.randint(lo, hi + 1)), room))
    total = min(total, max(room, n_seg))
    share = rs.dirichlet(np.full(n_seg, 2.0))
    counts = [max(3, int(round(x * total))) for x in share]
    while sum(counts) > max(room, 3 * n_seg):
        counts[int(np.argmax(counts))] -= 1
    segs = [(s0, G.text_ids(int(rs.randint(1 << 30)), 2) + [None] * (n - 2), min(e0, frames - 1))
            for s0, n, e0 in peaked_segments(counts, first_pos)]
    return G.window_script(ts0, eot, segs, "eot")


def ragged_island_windows(durations, seed=77, ts0=50364, eot=50257, peaked=False):
    """Per island (duration in seconds, a multiple of 30) one ragged window script per 30 s window."""
    import numpy as np
    rs = np.random.RandomState(seed)
    make = peaked_window if peaked else ragged_window
    return [[make(rs, 1500, ts0, eot) for _ in range(int(d) // 30)] for d in durations]


def script_ragged_islands(indices, durations=(), seed=77, peaked=False):
    """transcribe_many(streams=N, on_batch=functools.partial(script_ragged_islands, durations=...)): in a worker process,
    the ragged scripts of ITS islands (`indices` = positions in the caller's list, in stream order)."""
    from whisper_double.decoding import Script, set_row_scripts
    from whisper_timestamped import streams
    windows = ragged_island_windows(durations, seed, peaked=peaked)
    scripts = [Script(windows[i]) for i in indices]

    def on_group(rows):
        for r in rows:
            scripts[r].begin_window()
        set_row_scripts([scripts[r] for r in rows])
    streams.ON_GROUP_DECODE = on_group


def script_batch_cpu(indices):
    """CPU host-logic tests of transcribe_many(streams=N): in the worker process, the oracle-backed stand-ins for the kernels
    (incl. the B-stream entries) once, then the scripted transcript for every recording of `indices`."""
    global _STANDIN
    if not _STANDIN:
        import torch
        import cpu_kernel_standin
        from test_streams_host import install_streams_standin
        torch.set_num_threads(4)
        cpu_kernel_standin.install(_Patch())
        install_streams_standin(_Patch())
        _STANDIN = True
    script_batch(indices)
