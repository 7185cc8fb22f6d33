#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace, optional PMC)
into a plain-text per-kernel table (what `--stats` reports), for profiles/.

usage: rocpd_stats.py <results.db> [--skip N]   (skip the first N dispatches of each kernel = warm-up)
"""
import re
import sqlite3
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    db = sqlite3.connect(path)
    rows = db.execute(
        "select s.kernel_name, d.start, d.end, d.grid_size_x*d.grid_size_y*d.grid_size_z, d.workgroup_size_x, "
        "d.group_segment_size, s.arch_vgpr_count, s.sgpr_count from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    per = {}
    for name, st, en, grid, wg, lds, vg, sg in rows:
        per.setdefault(name, []).append((en - st, grid, wg, lds, vg, sg))
    tot = sum(sum(x[0] for x in v[skip:]) for v in per.values())
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s} "
          f"{'grid':>9s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'sgpr':>5s}")
    for name, v in sorted(per.items(), key=lambda kv: -sum(x[0] for x in kv[1][skip:])):
        d = [x[0] for x in v[skip:]] or [0]
        short = re.sub(r"\(.*", "", name)
        short = short if len(short) <= 70 else short[:67] + "..."
        print(f"{short:70s} {len(d):6d} {sum(d)/len(d)/1e3:10.2f} {min(d)/1e3:10.2f} {max(d)/1e3:10.2f} "
              f"{sum(d)/1e6:10.3f} {100*sum(d)/max(tot,1):6.1f} {v[-1][1]:9d} {v[-1][2]:5d} {v[-1][3]:7d} {v[-1][4]:5d} {v[-1][5]:5d}")
    try:
        # one row per (dispatch, counter, dimension instance): sum over the instances of a dispatch, then average
        pmc = db.execute("select kname, cname, avg(v), count(*) from (select s.kernel_name as kname, p.name as cname, "
                         "sum(e.value) as v from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                         "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s "
                         "on d.kernel_id = s.id group by d.id, p.name) group by kname, cname").fetchall()
        if pmc:
            print("\nPMC (sum over counter instances, average per dispatch)")
            for name, cname, val, cnt in pmc:
                print(f"{re.sub(r'[(].*', '', name)[:70]:70s} {cname:24s} {val:16.1f}  (dispatches={cnt})")
    except sqlite3.Error as e:
        print("no pmc:", e)


if __name__ == "__main__":
    main()
