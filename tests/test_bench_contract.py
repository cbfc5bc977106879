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
        assert r["kernel"] == "k_coop" and r["traffic_source"] and "committed profile" in r["traffic_source"]
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
    rows = [ln.replace("[early exit]", "").split() for ln in open(table) if "k_coop<" in ln]
    calls = sum(int(r[-6]) for r in rows)
    avg_ns = sum(float(r[-5]) for r in rows) / calls
    assert calls >= k["launches"]  # warm-up launches are in the table too
    assert abs(avg_ns * 1e-3 - k["usec"]) <= 0.10 * k["usec"]
    pmc = json.load(open(nodes[-1].replace("_bench_nodes_only.json", "_pmc_traffic.json")))["kernels"]
    tr = [v for kk, v in pmc.items() if kk.startswith("k_coop")][0]["traffic_bytes"]
    assert tr < 0.1 * k["bytes"]  # register-resident factor: HBM moves a few percent of the algorithmic bytes
    c5 = json.load(open(nodes[-1].replace("_bench_nodes_only.json", "_pmc_traffic_cfg5.json")))["kernels"]
    tail = [v for kk, v in c5.items() if kk.startswith("k_tail_fwd")][0]
    assert tail["early_exit_dispatches"] > 0 and 0.9e8 <= tail["traffic_bytes"] <= 1.2e8  # 12.5 M entries x 8 B
