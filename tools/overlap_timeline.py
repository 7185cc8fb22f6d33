#!/usr/bin/env python3
"""What do the batches in flight actually overlap?  From a rocprofv3 --kernel-trace database (rocpd sqlite) of the
pipelined kernel-level run: per kernel, the share of its own run time during which (a) any other kernel, (b) an
HBM-bound kernel (log-prob gather, rowmean, colnorm, logmel_finalize) was executing concurrently, plus the chip-level
picture: time with 0 / 1 / 2+ kernels resident and the busy span per step.

usage: overlap_timeline.py <results.db> [--skip-first-us N] [--name-filter substring]
"""
import re
import sqlite3
import sys

HBM_BOUND = ("logprob_gather_kernel", "rowmean_kernel", "rowmean_any_kernel", "colnorm_kernel", "logmel_finalize_kernel")


def short(name):
    s = re.sub(r"\(.*", "", name)
    s = re.sub(r"^void\s+", "", s)
    s = re.sub(r"^wt::", "", s)
    return s[:60]


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"d.{qcol}" if qcol else "0"
    rows = db.execute(f"select s.kernel_name, d.start, d.end, {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    rows = [(short(n), int(a), int(b), q) for n, a, b, q in rows if "wt::" in n or "wt_" in n or "kernel" in n]
    if not rows:
        print("no dispatches")
        return
    # keep the steady state: drop the first 40 % of the trace (workload construction, warm-up, single-stream pass unless filtered)
    frac = 0.0
    if "--tail" in sys.argv:
        frac = 1.0 - float(sys.argv[sys.argv.index("--tail") + 1])
    t_lo = rows[0][1] + frac * (rows[-1][2] - rows[0][1])
    rows = [r for r in rows if r[1] >= t_lo]
    n = len(rows)
    print(f"{n} dispatches, {len(set(r[3] for r in rows))} queue(s) [{qcol}], span {(rows[-1][2] - rows[0][1]) / 1e6:.3f} ms")

    # sweep: events (+1 / -1), to get the resident-kernel-count histogram
    ev = sorted([(a, 1) for _, a, b, _ in rows] + [(b, -1) for _, a, b, _ in rows])
    hist, cur, last = {}, 0, ev[0][0]
    for t, d in ev:
        hist[cur] = hist.get(cur, 0) + (t - last)
        last = t
        cur += d
    tot = sum(hist.values())
    print("resident kernels : share of the span   " + "   ".join(f"{k}: {100 * v / tot:.1f} %" for k, v in sorted(hist.items())))

    # per kernel: overlap with others.  O(n * window): dispatches are sorted by start; look back / ahead while intervals can intersect
    starts = [r[1] for r in rows]
    import bisect
    per = {}
    max_len = max(b - a for _, a, b, _ in rows)
    for i, (name, a, b, q) in enumerate(rows):
        lo = bisect.bisect_left(starts, a - max_len)
        hi = bisect.bisect_right(starts, b)
        any_iv, hbm_iv = [], []
        for j in range(lo, hi):
            if j == i:
                continue
            n2, a2, b2, q2 = rows[j]
            x, y = max(a, a2), min(b, b2)
            if y > x:
                any_iv.append((x, y))
                if n2.startswith(HBM_BOUND) or any(h in n2 for h in HBM_BOUND):
                    hbm_iv.append((x, y))

        def union(iv):
            iv.sort()
            s, end = 0, None
            for x, y in iv:
                if end is None or x > end:
                    s += y - x
                    end = y
                elif y > end:
                    s += y - end
                    end = y
            return s
        p = per.setdefault(name, [0, 0, 0, 0])
        p[0] += 1
        p[1] += b - a
        p[2] += union(any_iv)
        p[3] += union(hbm_iv)
    print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>9s} {'beside any kernel':>18s} {'beside an HBM-bound kernel':>27s}")
    for name, (c, t, o_any, o_hbm) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:60s} {c:6d} {t / c / 1e3:9.2f} {100 * o_any / t:17.1f}% {100 * o_hbm / t:26.1f}%")


if __name__ == "__main__":
    main()
