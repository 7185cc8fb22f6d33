#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC databases (FETCH_SIZE pass, WRITE_SIZE pass).

/opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read -> it is DOUBLED here; WRITE_SIZE is uncalibrated and taken as is.  Both counters are in KB.
usage: pmc_traffic.py <fetch.db> <write.db> [--workload NAME]
       -> JSON {"_workload": NAME, kernel: {fetch_bytes, write_bytes, hbm_bytes}}
bench.py only quotes a committed summary whose "_workload" equals the workload it is running.
"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select kname, avg(v), count(*) from (select s.kernel_name as kname, sum(e.value) as v from rocpd_pmc_event e "
        "join rocpd_info_pmc p on e.pmc_id = p.id join rocpd_kernel_dispatch d on e.event_id = d.event_id "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id where p.name = ? group by d.id) group by kname",
        (counter,)).fetchall()
    return {re.sub(r"\(.*", "", k): (v, n) for k, v, n in rows}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    if "--workload" in sys.argv:
        out["_workload"] = sys.argv[sys.argv.index("--workload") + 1]
    for k in sorted(set(fetch) | set(write)):
        if "wt" not in k:
            continue
        f = fetch.get(k, (0.0, 0))[0] * 1024.0 * 2.0     # KB -> bytes, x2 gfx950 correction
        w = write.get(k, (0.0, 0))[0] * 1024.0
        out[k] = dict(fetch_bytes=round(f), write_bytes=round(w), hbm_bytes=round(f + w),
                      dispatches=fetch.get(k, (0, 0))[1])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
