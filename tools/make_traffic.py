#!/usr/bin/env python
"""Build profiles/rNN_traffic.json from rocprofv3 PMC passes.

Usage: python tools/make_traffic.py OUT.json WORKLOAD=FETCH.db,WRITE.db [WORKLOAD=...]
Each .db is the rocpd database of a `rocprofv3 --pmc FETCH_SIZE` (or WRITE_SIZE) run of
`bench.py --workload WORKLOAD` (separate passes, kernel-trace only, as MI355X_MICROARCH.md's HBM section
prescribes).  Per kernel and launch: traffic = 2 * FETCH_SIZE KiB (the gfx950 correction of that guide for wide
coalesced reads) + WRITE_SIZE KiB."""
import json
import re
import sqlite3
import sys


def per_launch(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select k.name, count(distinct p.dispatch_id), sum(p.counter_value) from pmc_events p "
        "join kernels k on k.dispatch_id = p.dispatch_id where p.counter_name = ? group by 1", (counter,)).fetchall()
    out = {}
    for name, n, val in rows:
        short = re.sub(r"<.*", "", re.sub(r"\(.*", "", name).replace("void ", "")).replace("rnnt::", "")
        acc = out.setdefault(short, [0, 0.0])
        acc[0] += n
        acc[1] += val
    return {k: v[1] / max(v[0], 1) for k, v in out.items()}


def main():
    res = {"_comment": "HBM traffic per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                       "runs of bench.py per workload; summaries next to this file). FETCH_SIZE is doubled as "
                       "MI355X_MICROARCH.md (HBM section) prescribes for gfx950; WRITE_SIZE is taken as is. "
                       "Units: the counters are KiB."}
    for spec in sys.argv[2:]:
        wl, dbs = spec.split("=")
        fdb, wdb = dbs.split(",")
        f, w = per_launch(fdb, "FETCH_SIZE"), per_launch(wdb, "WRITE_SIZE")
        res[wl] = {k: {"FETCH_SIZE_KiB": round(f[k], 1), "WRITE_SIZE_KiB": round(w.get(k, 0.0), 1),
                       "traffic_bytes": int((2 * f[k] + w.get(k, 0.0)) * 1024)} for k in sorted(f)}
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
