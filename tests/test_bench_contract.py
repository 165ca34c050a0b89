"""The bench line's contract (driver prompt, section 4): checked on the line recorded by the last GPU run of this
round (profiles/rNN_bench_default.json) and on bench.py's source, without a GPU."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no recorded bench line under profiles/"
    with open(files[-1]) as fh:
        return json.loads(fh.readline())


def test_recorded_bench_line_has_every_contract_field():
    d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["unit"] == "ESS/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # achieved = algorithmic bytes per launch / the kernel's average launch duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.02 * r["achieved"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # the comparator is this repository's own port on a bounded sample: it must say so, and its rate must be consistent
    assert c.get("extrapolated") is True and "not rstan" in c["sample"]
    cells = 20000 * 200
    assert abs(c["ns_per_cell_per_thread"] - 1e9 * c["cores"] / (c["grad_evals_per_s"] * cells)) < 0.02 * c["ns_per_cell_per_thread"]
    # whole-job consistency: the gradient evaluations of the run at the algorithmic bytes each cannot exceed the HBM peak,
    # and the kernel time they imply cannot exceed the wall time
    cfg = d["config"]
    b_unit = r["algorithmic_bytes_per_launch"] / round(r["algorithmic_bytes_per_launch"] / 16964800.0)
    wall = d["ms_per_step"] * 1e-3 * d["steps"]
    assert cfg["grad_evals"] * b_unit / wall / 1e9 < r["peak"]
    chains = cfg["chains_total"]
    assert cfg["grad_evals"] / chains * r["avg_launch_ms"] * 1e-3 < wall


def test_rocprof_summary_agrees_with_the_bench_line():
    """The committed rocprofv3 --stats summary of the same command: the log-likelihood kernel's average duration there
    (all launches, including the shorter ones after chains have finished) must not exceed the HIP-event average of the
    bench line (launches with every chain active) and must be within 25 % of it."""
    import csv
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv")))
    assert files
    rows = list(csv.DictReader(open(files[-1])))
    lk = [r for r in rows if "ppcx_loglik_kernel" in r["Name"]]
    assert len(lk) == 1
    avg_ms = float(lk[0]["AverageNs"]) * 1e-6
    prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_under_rocprof.json")))
    with open(prof[-1]) as fh:
        ev_ms = json.loads(fh.readline())["roofline"]["avg_launch_ms"]
    assert 0.75 * ev_ms <= avg_ms <= 1.02 * ev_ms
    # the dominant kernel is the one the roofline object names
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert "ppcx_loglik_kernel" in top["Name"]
