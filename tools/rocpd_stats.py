#!/usr/bin/env python
"""Per-kernel summary (calls, total, avg, min, max, share) from a rocprofv3 rocpd sqlite output.

    python tools/rocpd_stats.py gpurun_out/<dir>/<name>_results.db [header text] > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = float(sum(r[2] for r in rows)) or 1.0
    for h in sys.argv[2:]:
        print("# " + h)
    print("%-72s %8s %13s %10s %8s %9s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
    for r in rows:
        name = r[0].replace("(anonymous namespace)::", "")
        print("%-72s %8d %13d %10.1f %8d %9d %6.2f" % (name[:72], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))


if __name__ == "__main__":
    main()
