#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 PMC databases (one or more passes).
usage: pmc_counters.py <pass1.db> [<pass2.db> ...]   -> lines "kernel  counter  avg-per-dispatch  (dispatches=n)" """
import re
import sqlite3
import sys


def main():
    rows_out = []
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        rows = db.execute(
            "select kname, cname, avg(v), count(*) from (select s.kernel_name as kname, p.name as cname, sum(e.value) as v "
            "from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on "
            "e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by d.id, p.name) "
            "group by kname, cname").fetchall()
        rows_out += [(re.sub(r"\(.*", "", k), c, v, n) for k, c, v, n in rows if "wt" in k]
    for k, c, v, n in sorted(rows_out):
        print(f"{k:70s} {c:24s} {v:16.1f}  (dispatches={n})")


if __name__ == "__main__":
    main()
