"""Debug figures of ``plot_word_alignment=True`` / ``"<prefix>"`` (the CLI's ``--plot``).

What the reference draws per aligned segment (/root/reference/whisper_timestamped/transcribe.py:1586-1646 the attention
map with the DTW path over word rows and the log-mel below it, :1685-1700 the per-token attention curves with the peaks
the disfluency rule looks at, :1756-1781 the word boundaries, then ``<prefix>.alignment<NNN>.jpg`` or ``plt.show()``)
and per recording when ``vad`` is on (:2139-2150 the waveform with the speech islands, ``<prefix>.VAD.jpg``).

Everything drawn here is data the alignment already has: the unit's cost matrix and warping path are copied from the
device once the batch's words exist (``AlignmentBatch(plot=...)`` keeps them); the peak positions of the third panel are
recomputed on the host with the reference's own ``scipy.signal.find_peaks`` call -- a figure, not a result.
matplotlib is imported when the first figure is drawn.
"""
from __future__ import annotations

import numpy as np

from .words import AUDIO_TIME_PER_TOKEN

_count = 0          # alignments of the current transcribe() call (the reference's num_alignment_for_plot, :300-301, :1583-1584)


def reset():
    global _count
    _count = 0


def _finish(fig, plot, suffix):
    import matplotlib.pyplot as plt
    if isinstance(plot, str):
        fig.savefig(f"{plot}.{suffix}.jpg", bbox_inches="tight", pad_inches=0)
        plt.close(fig)
    else:
        plt.show()


def _word_rows(unit):
    """[(label, first token row, one past the last)] of every word of the unit, timestamp words included."""
    rows, r = [], 0
    for pieces in unit.word_pieces:
        rows.append(("|".join(pieces).strip(), r, r + len(pieces)))
        r += len(pieces)
    return rows


def alignment_figure(unit, cost, path_tokens, path_frames, jumps, words, plot):
    """One figure for one aligned unit.  cost: (T, F) as the DTW saw it (negated attention); path_*: the warping path;
    jumps: int[T + 1]; words: what finish_unit returned for the unit (absolute times)."""
    import matplotlib.pyplot as plt
    global _count
    _count += 1
    attention = -np.asarray(cost, dtype=np.float64)
    T, F = attention.shape
    mel = getattr(unit, "mel", None)
    panels = ["attention"] + (["mel"] if mel is not None else []) + (["peaks"] if unit.detect_disfluencies else [])
    fig, axes = plt.subplots(len(panels), 1, figsize=(16, 9), squeeze=False,
                             gridspec_kw={"height_ratios": [3] + [1] * (len(panels) - 1)})
    axes = dict(zip(panels, axes[:, 0]))
    t0 = unit.start_token * AUDIO_TIME_PER_TOKEN
    per_second = int(round(1 / AUDIO_TIME_PER_TOKEN))
    second_ticks = np.arange(0, F, per_second)
    second_labels = [f"{t0 + x * AUDIO_TIME_PER_TOKEN:.2f}" for x in second_ticks]

    # spans of the returned words in frames of this window; a disfluency mark carries no token
    spans = []
    for w in words:
        b, e = (w["start"] - t0) / AUDIO_TIME_PER_TOKEN, (w["end"] - t0) / AUDIO_TIME_PER_TOKEN
        idx = list(w.get("tokens_indices") or [])
        spans.append((w["text"], b, e, (min(idx) if idx else None)))

    ax = axes["attention"]
    ax.imshow(attention, aspect="auto", interpolation="nearest")
    ax.plot(np.asarray(path_frames), np.asarray(path_tokens), color="red", linewidth=1.0)
    rows = _word_rows(unit)
    for _, _, stop in rows:
        ax.axhline(stop - 0.5, color="black", linestyle="dashed", linewidth=0.8)
    ax.set_yticks([(a + b) / 2 - 0.5 for _, a, b in rows])
    ax.set_yticklabels([label for label, _, _ in rows])
    ax.tick_params(axis="y", length=0)
    ax.set_ylabel("Words")
    ax.set_ylim(T - 0.5, -0.5)
    for text, b, e, top in spans:
        if top is None:                                   # "[*]": shade what the disfluency rule cut away
            ax.axvspan(b, e, color="orange", alpha=0.15)
            continue
        for x in (b, e):
            ax.plot([x, x], [top - 0.5, T - 0.5], color="red", linestyle="dotted", linewidth=1.0)
        if mel is None:
            ax.text(b, T - 0.5, text, ha="left", va="bottom", color="red")
    ax.set_xlim(-0.5, F - 0.5)
    ax.set_xticks(second_ticks)
    last = panels[-1]
    if last != "attention":
        ax.set_xticklabels([])

    if mel is not None:
        ax = axes["mel"]
        m = np.asarray(mel.detach().float().cpu() if hasattr(mel, "detach") else mel)
        m = m.reshape(-1, m.shape[-1])[:, 2 * unit.start_token:2 * unit.end_token]       # 10 ms columns: two per frame
        ax.imshow(m, aspect="auto", origin="lower", interpolation="nearest")
        ax.set_yticks([])
        ax.set_ylabel("MFCC")
        for text, b, e, top in spans:
            if top is None:
                continue
            ax.text(2 * b, m.shape[0] * 1.05, text, ha="left", va="bottom", color="red", clip_on=False)
            for x in (b, e):
                ax.axvline(2 * x, color="red", linestyle="dotted", linewidth=1.0)
        ax.set_xlim(-0.5, 2 * F - 0.5)
        ax.set_xticks(2 * second_ticks)
        if last != "mel":
            ax.set_xticklabels([])

    if unit.detect_disfluencies:
        from scipy.signal import find_peaks
        ax = axes["peaks"]
        for i in range(T):
            begin, end = int(jumps[i]), int(jumps[i + 1])
            if end <= begin:
                continue
            curve = attention[i, begin:end]
            ax.plot(np.arange(begin, end), curve)
            peaks, props = find_peaks(curve, width=3, prominence=0.02)
            several = len(peaks) > 1
            for k, p in enumerate(peaks):
                ax.axvline(begin + p, color="red" if (several and k < len(peaks) - 1) else "green", linestyle="--", linewidth=0.8)
            if len(peaks):
                left, right = begin + props["left_bases"], begin + props["right_bases"]
                ax.bar((left + right) / 2, props["prominences"], width=right - left, alpha=0.5,
                       color="red" if several else "green")
        ax.set_xlim(-0.5, F - 0.5)
        ax.set_xticks(second_ticks)

    scale = 2 if last == "mel" else 1
    axes[last].set_xticks(scale * second_ticks)
    axes[last].set_xticklabels(second_labels)
    axes[last].set_xlabel("Time (s)")
    _finish(fig, plot, f"alignment{_count:03d}")


def vad_figure(audio, islands_in_samples, sample_rate, plot):
    """The waveform (at most ~10 000 points) with the speech islands shaded: transcribe.py:2139-2150."""
    import matplotlib.pyplot as plt
    x = np.asarray(audio.detach().cpu() if hasattr(audio, "detach") else audio).reshape(-1)
    step = x.shape[0] // 10000 + 1
    fig, ax = plt.subplots()
    ax.plot(np.arange(0, x.shape[0], step) / sample_rate, x[::step])
    for s, e in islands_in_samples:
        ax.axvspan(s / sample_rate, e / sample_rate, color="red", alpha=0.1)
    _finish(fig, plot, "VAD")
