#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (``*_results.db``) into a small markdown table:
per-kernel calls / total / average / min / max duration (and PMC counter sums when present).
Usage: python tools/rocpd_summary.py <results.db> [title] > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)            # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    print("# %s\n" % title)
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |"
              % (short(name), n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    try:
        pmc = db.execute(
            "select k.name, p.counter_name, count(*), sum(p.counter_value) from pmc_events p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by 1, 2 order by 1, 2").fetchall()
    except sqlite3.Error:
        pmc = []
    if pmc:
        print("\n| kernel | counter | dispatches | sum | per dispatch |")
        print("|---|---|---|---|---|")
        for name, cname, n, val in pmc:
            print("| `%s` | %s | %d | %.6g | %.6g |" % (short(name), cname, n, val, val / max(n, 1)))


if __name__ == "__main__":
    main()
