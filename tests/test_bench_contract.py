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
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
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
    table = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_rocprofv3_kernel_stats.txt")))[-1]
    rows = [ln for ln in open(table) if name + "<" in ln or ln.startswith(name + "(") or (" " + name + "(") in ln]
    assert rows, (name, table)
    avg_ns = float(rows[0].split()[-4])
    assert abs(avg_ns * 1e-3 - d["roofline"]["usec_per_launch"]) <= 0.10 * d["roofline"]["usec_per_launch"]
