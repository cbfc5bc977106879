#!/usr/bin/env python
"""Mean PMC counter value per kernel from a rocprofv3 rocpd sqlite output (JSON on stdout).

    python tools/rocpd_pmc.py <results.db>

Dispatches that returned at once (see tools/rocpd_stats.py: launches queued behind a decided test) move no
data; they are counted separately ("early_exit_dispatches") and left out of the mean: a dispatch is one of
them when its counter value is below 30 % of the kernel's median and the kernel's smallest value is.
"""
import json
import re
import sqlite3
import sys

CHUNK_KERNELS = re.compile(r"(k_panel_|k_tail_|k_fold_|k_check_)")  # see tools/rocpd_stats.py


def main():
    con = sqlite3.connect(sys.argv[1])
    per = {}
    for name, ctr, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
        name = name.replace("(anonymous namespace)::", "")
        name = name.split("(")[0].replace("void ", "").strip()
        per.setdefault((name, ctr), []).append(value)
    out = {}
    for (name, ctr), vs in per.items():
        vs.sort()
        med = vs[len(vs) // 2]
        early = 0
        if CHUNK_KERNELS.search(name) and med > 0 and vs[0] < 0.3 * med:
            early = sum(1 for v in vs if v < 0.3 * med)
            vs = [v for v in vs if v >= 0.3 * med]
        out.setdefault(name, {})[ctr] = dict(dispatches=len(vs), mean=sum(vs) / float(len(vs)), min=vs[0], max=vs[-1],
                                            early_exit_dispatches=early)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
