#!/usr/bin/env python
"""Mean PMC counter value per kernel from a rocprofv3 rocpd sqlite output (JSON on stdout).

    python tools/rocpd_pmc.py <results.db>
"""
import json
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                       "from counters_collection group by kernel_name, counter_name").fetchall()
    out = {}
    for name, ctr, cnt, avg, mn, mx in rows:
        name = name.replace("(anonymous namespace)::", "")
        name = name.split("(")[0].replace("void ", "").strip()
        out.setdefault(name, {})[ctr] = dict(dispatches=cnt, mean=avg, min=mn, max=mx)
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
