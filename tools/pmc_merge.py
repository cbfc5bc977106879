#!/usr/bin/env python
"""Merge the FETCH_SIZE and WRITE_SIZE passes (tools/rocpd_pmc.py output) into profiles/rNN_pmc_traffic.json.

    python tools/pmc_merge.py fetch.json write.json "<command line profiled>" [bench.json] > profiles/r01_pmc_traffic.json

bench.json (optional): the line bench.py printed in the FETCH pass.  A launch of the resident search grid (k_coop_run) is as
long as the call it serves, so a mean per dispatch says little: with the record's count of nodes over all launches of the
process the kernel also gets `traffic_bytes_per_node`.

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads exactly half of the bytes of a wide
coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.  Both counters are in KiB.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) -- "
                     + sys.argv[3] + "; MI355X; summarised on the GPU box by tools/rocpd_pmc.py + tools/pmc_merge.py",
           "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch (mean over all dispatches of the kernel)",
           "correction": "gfx950: FETCH_SIZE reads exactly 1/2 of the bytes of a wide coalesced streaming read "
                         "(MI355X_MICROARCH.md, HBM section) -> fetch is doubled; WRITE_SIZE is taken as is",
           "kernels": {}}
    try:  # which device code the counters belong to (bench.py refuses a summary of other code)
        from miosqp_amd import _lib
        out["source_digest"] = _lib.source_digest()
    except Exception:
        pass
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, {}).get("FETCH_SIZE")
        w = write.get(k, {}).get("WRITE_SIZE")
        rec = {"dispatches": (f or w)["dispatches"], "early_exit_dispatches": (f or w).get("early_exit_dispatches", 0)}
        if f:
            rec["FETCH_SIZE_KiB"] = round(f["mean"], 3)
        if w:
            rec["WRITE_SIZE_KiB"] = round(w["mean"], 3)
        rec["traffic_bytes"] = int(round(1024 * (2 * (f["mean"] if f else 0.0) + (w["mean"] if w else 0.0))))
        out["kernels"][k] = rec
    if len(sys.argv) > 4:
        try:
            line = [ln for ln in open(sys.argv[4]).read().splitlines() if ln.startswith("{")][-1]
            bench = json.loads(line)
            for kr in bench["roofline"]["kernels"]:
                allk = kr.get("all_launches_of_the_process")
                if not allk:
                    continue
                for name, rec in out["kernels"].items():
                    if name == kr["kernel"] or name.startswith(kr["kernel"] + "<"):
                        rec["nodes_of_all_dispatches"] = allk["nodes"]
                        rec["traffic_bytes_per_node"] = int(round(rec["traffic_bytes"] * rec["dispatches"] / max(1, allk["nodes"])))
        except Exception as ex:  # noqa: BLE001
            out["bench_record_error"] = repr(ex)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
