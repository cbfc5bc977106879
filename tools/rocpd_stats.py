#!/usr/bin/env python
"""Per-kernel summary (calls, total, avg, min, max, share) from a rocprofv3 rocpd sqlite output.

    python tools/rocpd_stats.py gpurun_out/<dir>/<name>_results.db [header text] > profiles/<name>.txt

Launches that returned at once are listed on their own line ("<kernel> [early exit]"): the multi-kernel
forms keep one chunk of launches queued ahead of the host's decision, and every kernel of a chunk queued
behind a decided test returns on `ctrl->done` after ~1.5 us.  A kernel's dispatches are split when its
shortest one is below 30 % of its median; the threshold is 30 % of the median.  Only the kernels of such chunks
(k_panel_*, k_tail_*, k_fold_*, k_check_*) are split: a short k_coop launch is a node with few iterations.
"""
import re
import sqlite3
import sys

# only the kernels of a queued-ahead chunk can return at once; a short k_coop launch is a short node, not an early exit
CHUNK_KERNELS = re.compile(r"(k_panel_|k_tail_|k_fold_|k_check_)")


def main():
    con = sqlite3.connect(sys.argv[1])
    per = {}
    for name, dur in con.execute("select name, duration from kernels"):
        per.setdefault(name, []).append(dur)
    rows = []
    for name, ds in per.items():
        ds.sort()
        med = ds[len(ds) // 2]
        if CHUNK_KERNELS.search(name) and ds[0] < 0.3 * med:
            full = [d for d in ds if d >= 0.3 * med]
            early = [d for d in ds if d < 0.3 * med]
            rows.append((name, full))
            rows.append((name + " [early exit]", early))
        else:
            rows.append((name, ds))
    rows.sort(key=lambda r: -sum(r[1]))
    tot = float(sum(sum(r[1]) for r in rows)) or 1.0
    for h in sys.argv[2:]:
        print("# " + h)
    print("%-72s %8s %13s %10s %8s %9s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "%"))
    for name, ds in rows:
        name = name.replace("(anonymous namespace)::", "")
        print("%-72s %8d %13d %10.1f %8d %9d %6.2f" % (name[:72], len(ds), sum(ds), sum(ds) / float(len(ds)), ds[0],
                                                        ds[-1], 100 * sum(ds) / tot))


if __name__ == "__main__":
    main()
