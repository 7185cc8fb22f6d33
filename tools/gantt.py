#!/usr/bin/env python3
"""Gantt of the pipelined region from a rocprofv3 --kernel-trace database (rocpd sqlite): the dispatches of the longest
gap-free stretch near the end of the trace (one timed region of the kernel-level leg), one line per dispatch:
start / end in microseconds from the stretch's first dispatch, queue, kernel; then the time during which NO HBM-bound kernel
(log-prob gather, rowmean, colnorm, logmel_finalize) was executing, and the list of those holes.

usage: gantt.py <results.db> [--steps N]   (N = how many log-prob gathers to print, default 6)
"""
import re
import sqlite3
import sys

HBM_BOUND = ("logprob_gather_kernel", "rowmean_kernel", "rowmean_any_kernel", "colnorm_kernel", "logmel_finalize_kernel")


KNOWN = ("logprob_gather_kernel", "logprob_digest_kernel", "rowmean_kernel", "rowmean_any_kernel", "colnorm_kernel", "fix00_kernel",
         "dtw_kernel", "small_tail_kernel", "stft_mel_kernel", "logmel_finalize_kernel", "padding_after_finalize_kernel",
         "logmel_init_kernel", "find_start_padding_kernel")


def short(name):
    for k in KNOWN:          # (the database holds mangled names: _ZN2wt21logprob_gather_kernelIfEE...)
        if k in name:
            return k
    return name[:28]


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 6
    cols = [r[1] for r in db.execute("pragma table_info(rocpd_kernel_dispatch)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"d.{qcol}" if qcol else "0"
    rows = db.execute(f"select s.kernel_name, d.start, d.end, {sel} from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    rows = [(short(n), int(a), int(b), q) for n, a, b, q in rows if "wt::" in n or "_ZN2wt" in n]
    # stretches without a gap > 200 us between consecutive starts; take the last one with >= 200 dispatches
    stretches, cur = [], [rows[0]]
    for r in rows[1:]:
        if r[1] - max(x[2] for x in cur[-8:]) > 200_000:
            stretches.append(cur)
            cur = []
        cur.append(r)
    stretches.append(cur)
    big = [s for s in stretches if len(s) >= 100]
    st = big[-1]
    # the middle of the stretch
    gathers = [i for i, r in enumerate(st) if r[0].startswith("logprob_gather")]
    mid = len(gathers) // 2
    i0, i1 = gathers[mid], gathers[min(mid + steps, len(gathers) - 1)]
    part = st[i0:i1 + 1]
    t0 = part[0][1]
    qs = sorted(set(r[3] for r in part))
    print(f"stretch of {len(st)} dispatches, {len(gathers)} gathers, {(st[-1][2] - st[0][1]) / 1e3 / max(1, len(gathers)):.1f} us per step; "
          f"printing {len(part)} dispatches, queues {qs}")
    for n, a, b, q in part:
        print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:7.1f}  q{qs.index(q)}  {n}")
    # holes: no HBM-bound kernel executing
    iv = sorted((a, b) for n, a, b, q in st if n.startswith(HBM_BOUND))
    holes, end = [], iv[0][1]
    for a, b in iv[1:]:
        if a > end:
            holes.append((end, a))
        end = max(end, b)
    tot = sum(b - a for a, b in holes)
    span = iv[-1][1] - iv[0][0]
    print(f"no HBM-bound kernel executing: {tot / 1e3:.1f} us of {span / 1e3:.1f} us ({100 * tot / span:.1f} %), {len(holes)} holes, "
          f"{tot / 1e3 / max(1, len(gathers)):.1f} us per step; sum of HBM-bound kernel times per step "
          f"{sum(b - a for a, b in iv) / 1e3 / max(1, len(gathers)):.1f} us")
    # which kernels run during holes
    during = {}
    for ha, hb in holes:
        for n, a, b, q in st:
            x, y = max(a, ha), min(b, hb)
            if y > x:
                during[n] = during.get(n, 0) + (y - x)
    print("during the holes: " + ", ".join(f"{n} {v / 1e3 / max(1, len(gathers)):.1f} us/step" for n, v in sorted(during.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
