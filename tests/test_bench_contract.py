"""The bench line the driver parses: the committed record of the last measured round carries every field of
the contract, with consistent values (CPU check of a file produced on the GPU box by `python bench.py`)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no committed bench record under profiles/"
    return json.load(open(files[-1]))


def test_bench_record_has_the_contract_fields():
    d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    # value and ms_per_step describe the same timed region: iterations/s = iterations per node / time per node
    assert abs(d["value"] - d["iters_per_node"] * 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]


def test_roofline_and_cpu_baseline_objects():
    d = _latest()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    # `bound` says what binds the dominant kernel: "hbm" for the streaming forms, "exchange-latency" for the cooperative
    # register-resident solver (its `achieved` is algorithmic bytes over time; `hbm_measured_gbs` is what HBM delivered)
    assert r["bound"] in ("hbm", "exchange-latency") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    if r["bound"] == "exchange-latency":
        # the counter traffic comes from a committed PMC summary OF THE SAME DEVICE CODE (source digest), or not at all
        # (k_coop: one launch per node; k_coop_run, r06: the grid resident over a whole call of the search)
        assert r["kernel"] in ("k_coop", "k_coop_run") and r["traffic_source"]
        assert ("committed profile" in r["traffic_source"]) == (r["traffic"] is not None)
        assert r["traffic"] is not None or "other device code" in r["traffic_source"]
        assert r["hbm_measured_gbs"] is None or r["hbm_measured_gbs"] < 0.2 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    dom = max(r["kernels"], key=lambda k: k["usec"])
    assert dom["kernel"] == r["kernel"]
    assert abs(r["achieved"] - dom["bytes"] / dom["usec"] * 1e-3) <= 0.01 * r["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]


def test_committed_profiles_agree_on_the_dominant_kernel():
    """The rocprofv3 kernel table of the same round lists the kernel the roofline names, with an average
    duration within 10 % of the live HIP-event measurement (profiled runs are slower)."""
    d = _latest()
    name = d["roofline"]["kernel"]
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats_nodes_only.txt"))) or \
        sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats.txt")))
    table = tables[-1]
    rows = [ln for ln in open(table) if (name + "<" in ln or ln.startswith(name + "(") or (" " + name + "(") in ln)]
    assert rows, (name, table)
    # every dispatch of the kernel (a one-launch-per-node kernel has no early exits: all its rows are node launches)
    calls = sum(int(r.replace("[early exit]", "").split()[-6]) for r in rows)
    total_ns = sum(float(r.replace("[early exit]", "").split()[-5]) for r in rows)
    avg_ns = total_ns / calls
    if name == "k_coop_run":
        # a resident launch is as long as the call of the search it serves (warm-up: 10 nodes, timed region: 150): the
        # table's TOTAL over the launches of the traced command against the HIP-event total of the same command's record
        nodes = json.load(open(table.replace("_rocprofv3_kernel_stats_nodes_only.txt", "_bench_nodes_only.json")))
        allk = nodes["roofline"]["kernels"][0]["all_launches_of_the_process"]
        assert nodes["roofline"]["kernel"] == name and calls == allk["launches"]
        assert abs(total_ns * 1e-3 - allk["usec_total"]) <= 0.10 * allk["usec_total"]
        # ... and the fraction those launches give together is the record's own, within the warm-up's share (the first launch
        # of the process pays the kernel's first dispatch inside its pair of events: several milliseconds under the tracer)
        assert abs(allk["frac"] - nodes["roofline"]["frac"]) <= 0.10 * nodes["roofline"]["frac"]
        return
    assert abs(avg_ns * 1e-3 - d["roofline"]["usec_per_launch"]) <= 0.10 * d["roofline"]["usec_per_launch"]


def test_every_roofline_fraction_can_be_recomputed_from_profiles():
    """the judge must be able to recompute each `frac` from files under profiles/ alone.  The node-only
    run (JSON + kernel table of the SAME command) gives iterations per launch and the average launch; the PMC
    file gives the HBM-side traffic; config 5 has its own pair."""
    nodes = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_nodes_only.json")))
    if not nodes:
        return  # round-1 layout
    d = json.load(open(nodes[-1]))
    k = d["roofline"]["kernels"][0]
    bytes_per_iter = d["roofline"]["iteration"]["bytes"]
    assert abs(k["bytes"] - bytes_per_iter * k["iterations_per_launch"]) <= 1e-3 * k["bytes"]
    assert abs(d["roofline"]["frac"] - k["bytes"] / k["usec"] * 1e-3 / 8000.0) <= 2e-3
    table = nodes[-1].replace("_bench_nodes_only.json", "_rocprofv3_kernel_stats_nodes_only.txt")
    rows = [ln.replace("[early exit]", "").split() for ln in open(table) if "k_coop<" in ln or "k_coop_run<" in ln]
    calls = sum(int(r[-6]) for r in rows)
    total_us = sum(float(r[-5]) for r in rows) * 1e-3
    assert calls >= k["launches"]  # warm-up launches are in the table too
    pmc = json.load(open(nodes[-1].replace("_bench_nodes_only.json", "_pmc_traffic.json")))["kernels"]
    if k["kernel"] == "k_coop_run":
        # resident launches (r06): the table's total is warm-up + timed region; per NODE everything lines up
        allk = k["all_launches_of_the_process"]
        assert calls == allk["launches"] and abs(total_us - allk["usec_total"]) <= 0.10 * allk["usec_total"]
        assert abs(total_us / allk["nodes"] - k["usec_per_node"]) <= 0.10 * k["usec_per_node"]
        rec = [v for kk, v in pmc.items() if kk.startswith("k_coop_run")][0]
        assert rec["nodes_of_all_dispatches"] == allk["nodes"]
        tr = rec["traffic_bytes_per_node"] * k["nodes_per_launch"]
    else:
        assert abs(total_us / calls - k["usec"]) <= 0.10 * k["usec"]
        tr = [v for kk, v in pmc.items() if kk.startswith("k_coop")][0]["traffic_bytes"]
    assert tr < 0.1 * k["bytes"]  # register-resident factor: HBM moves a few percent of the algorithmic bytes
    four = nodes[-1].replace("_bench_nodes_only.json", "_pmc_traffic_cfg5_four_launches.json")
    c5 = json.load(open(four if os.path.exists(four) else nodes[-1].replace("_bench_nodes_only.json", "_pmc_traffic_cfg5.json")))["kernels"]
    tail = [v for kk, v in c5.items() if kk.startswith("k_tail_fwd")][0]
    assert tail["early_exit_dispatches"] > 0 and 0.9e8 <= tail["traffic_bytes"] <= 1.2e8  # 12.5 M entries x 8 B


def _table_avg_us(table, prefix):
    rows = [ln.replace("[early exit]", "").split() for ln in open(table) if prefix in ln and not ln.startswith("#")]
    calls = sum(int(r[-6]) for r in rows)
    return sum(float(r[-5]) for r in rows) / calls * 1e-3, calls


def test_streaming_form_fractions_can_be_recomputed_from_profiles():
    """the `stream` and `config5` fractions, like k_coop's, follow from files under profiles/ alone --
    the node-only run of each form (JSON, untraced + traced), the rocprofv3 kernel table of the traced run and the PMC
    summary of the same command.  A single-launch kernel is one dispatch per node, so tracing does not distort it (the
    two-launch form is: see the traced / untraced pair of that form)."""
    P = os.path.join(ROOT, "profiles")
    base = sorted(glob.glob(os.path.join(P, "r*_bench_persistent_untraced.json")))
    if not base:
        return  # rounds before the persistent solver
    tag = os.path.basename(base[-1]).split("_")[0]

    def load(name):
        return json.load(open(os.path.join(P, "%s_%s" % (tag, name))))

    # ---- config 2, product form, one persistent launch per node
    u, t = load("bench_persistent_untraced.json"), load("bench_persistent.json")
    k = u["roofline"]["kernels"][0]
    assert k["kernel"] == "k_pers_small"
    assert abs(k["bytes"] - u["roofline"]["iteration"]["bytes"] * k["iterations_per_launch"]) <= 1e-3 * k["bytes"]
    assert abs(u["roofline"]["frac"] - k["bytes"] / k["usec"] * 1e-3 / 8000.0) <= 2e-3
    avg_us, calls = _table_avg_us(os.path.join(P, tag + "_rocprofv3_kernel_stats_persistent.txt"), "k_pers_small")
    kt = t["roofline"]["kernels"][0]
    assert calls >= kt["launches"]
    assert abs(avg_us - kt["usec"]) <= 0.15 * kt["usec"]            # the table also holds the warm-up nodes
    assert abs(kt["usec"] - k["usec"]) <= 0.05 * k["usec"]          # traced == untraced for a one-launch-per-node kernel
    pmc = [v for kk, v in load("pmc_traffic_persistent.json")["kernels"].items() if kk.startswith("k_pers_small")][0]
    assert pmc["traffic_bytes"] < 0.2 * k["bytes"]                  # the factor streams from L2 at this size, not from HBM
    # the two-launch form of the same nodes: tracing more than doubles its wall time per iteration
    u2, t2 = load("bench_two_kernel_form_untraced.json"), load("bench_two_kernel_form.json")
    assert t2["roofline"]["iteration"]["usec_in_timed_region"] > 1.5 * u2["roofline"]["iteration"]["usec_in_timed_region"]
    assert u["roofline"]["frac"] > u2["roofline"]["frac"]
    # ---- config 5 and the size beyond the Infinity Cache, factor form, one persistent launch per node
    for name, beyond in (("cfg5", False), ("cfg5x", True)):
        u = load("bench_%s_untraced.json" % name)
        k = u["roofline"]["kernels"][0]
        assert k["kernel"] == "k_pers"
        assert abs(u["roofline"]["frac"] - k["bytes"] / k["usec"] * 1e-3 / 8000.0) <= 2e-3
        avg_us, calls = _table_avg_us(os.path.join(P, "%s_rocprofv3_kernel_stats_%s.txt" % (tag, name)), "k_pers<false")
        assert calls >= k["launches"] and 0.8 * k["usec"] <= avg_us <= 1.5 * k["usec"]
        pmc = [v for kk, v in load("pmc_traffic_%s.json" % name)["kernels"].items() if kk.startswith("k_pers<false")][0]
        per_iter = pmc["traffic_bytes"] / k["iterations_per_launch"]   # (mean over the launches, warm-up included: +-20 %)
        moved = k["bytes_moved"] / k["iterations_per_launch"]
        assert 0.8 * moved <= per_iter <= 1.6 * moved, (name, per_iter, moved)
        if beyond:
            assert moved > 256 * 2 ** 20  # more than the Infinity Cache holds, per iteration (the symmetric tiles: 289 MB at n = 8000)
