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
    assert c["kind"] in ("reference", "port", "port-optimised") and c["cores"] >= 1 and c["value"] > 0
    if c["kind"] == "port-optimised":
        # both comparators are on the line; the optimised one (the product's formulation on the host) is the faster, agrees
        # with the literal port, and each sample stayed within its share of --cpu-seconds (default 15 s: 7.5 s each, + one call)
        lit, opt = c["port"], c["port_optimised"]
        assert opt["grad_evals_per_s"] > 5 * lit["grad_evals_per_s"] and opt["value"] == c["value"]
        assert opt["agrees_with_port"]["lp_rel"] < 1e-11 and opt["agrees_with_port"]["grad_rel_max"] < 1e-9
        assert lit["seconds_sampled"] < 12 and opt["seconds_sampled"] < 12
    # the comparator is this repository's own port on a bounded sample: it must say so, and its rate must be consistent
    assert c.get("extrapolated") is True and "not rstan" in c["sample"]
    # ... and beside the extrapolated cfg3 figure the line carries ONE measured pair: whole fits of cfg2 on the host's cores and
    # on the GPU, same seeds and estimator (round 5: part of the default run)
    if int(os.path.basename(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))[-1])[1:3]) >= 5:
        ms = c["measured"]
        assert ms["config"] == "cfg2" and ms["extrapolated"] is False and ms["cores"] >= 1
        assert ms["cpu_s"] > ms["gpu_s"] > 0 and ms["cpu_ess_per_s"] > 0 and ms["gpu_ess_per_s"] > ms["cpu_ess_per_s"]
        assert abs(ms["gpu_over_cpu_ess_per_s"] - ms["gpu_ess_per_s"] / ms["cpu_ess_per_s"]) < 0.06 * ms["gpu_over_cpu_ess_per_s"]
    cells = 20000 * 200
    assert abs(c["ns_per_cell_per_thread"] - 1e9 * c["cores"] / (c["grad_evals_per_s"] * cells)) < 0.02 * c["ns_per_cell_per_thread"]
    # whole-job consistency: the gradient evaluations of the run at the algorithmic bytes each cannot exceed the HBM peak,
    # and the kernel time they imply cannot exceed the wall time
    cfg = d["config"]
    b_unit = r["algorithmic_bytes_per_launch"] / round(r["algorithmic_bytes_per_launch"] / 16964800.0)
    wall = d["ms_per_step"] * 1e-3 * d["steps"]
    assert cfg["grad_evals"] * b_unit / wall / 1e9 < r["peak"]
    chains = cfg["chains_total"]
    if "pipelined" not in cfg.get("round_structure", ""):
        # three-launch rounds on one stream: the log-likelihood launches alone fit into the wall time (with chain groups on
        # several streams the launches of different groups overlap, and each covers a part of the chains)
        assert cfg["grad_evals"] / chains * r["avg_launch_ms"] * 1e-3 < wall
    # the roofline sample names where it was taken, and the single-stream fit it comes from is on the line
    if r["kernel"] == "ppcx_ls_kernel":
        assert "one in-order stream" in r["sampled_in"] and d["single_stream"]["stream_groups"] == 1
        assert abs(d["single_stream"]["kernel_ms"]["loglik_ms"] - r["avg_launch_ms"]) < 1e-6
    # the posterior-predictive kernel's object (BASELINE config 4: "posterior-predictive draw kernel, roofline report")
    p = d["ppc"]
    for k in ("kernel", "nb_draws", "kernel_ms", "nb_draws_per_s", "algorithmic_bytes_per_posterior_draw", "achieved", "peak", "frac", "bound"):
        assert k in p, k
    assert p["kernel"] in ("ppcx_ppc_kernel", "ppcx_ppc_table_kernel + ppcx_ppc_wave_kernel", "ppcx_ppc_table_kernel + ppcx_ppc_kernel") and p["bound"] in ("alu", "hbm")
    assert abs(p["nb_draws_per_s"] - p["nb_draws"] / (p["kernel_ms"] * 1e-3)) < 0.01 * p["nb_draws_per_s"]
    assert abs(p["frac"] - p["achieved"] / p["peak"]) < 1e-5


def test_rocprof_summary_agrees_with_the_bench_line():
    """The committed rocprofv3 --stats summary of the same command (one in-order stream: every launch of the dominant
    kernel then covers all chains): that kernel's average duration there (all launches, including the shorter ones after
    chains have finished) must not exceed the HIP-event average of the bench line (launches with every chain active) and
    must be within 25 % of it; and the posterior-predictive kernel has its row."""
    import csv
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv")))
    assert files
    rows = list(csv.DictReader(open(files[-1])))
    prof = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_under_rocprof.json")))
    with open(prof[-1]) as fh:
        line = json.loads(fh.readline())
    kernel = line["roofline"]["kernel"]
    lk = [r for r in rows if kernel in r["Name"]]
    assert len(lk) == 1
    avg_ms = float(lk[0]["AverageNs"]) * 1e-6
    ev_ms = line["roofline"]["avg_launch_ms"]
    assert 0.75 * ev_ms <= avg_ms <= 1.02 * ev_ms
    # the dominant kernel is the one the roofline object names
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert kernel in top["Name"]
    if int(os.path.basename(files[-1])[1:3]) >= 3:
        assert any("ppcx_ppc_kernel" in r["Name"] or "ppcx_ppc_wave_kernel" in r["Name"] for r in rows)


def test_recorded_config_lines_parse():
    """Every recorded bench / config line under profiles/ is one JSON object (a capture once kept RCCL's banner instead of
    the line): cfg4 gene-shard runs, cfg5 two-pass runs, ADVI timings."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_*.json")))
    assert files
    for f in files:
        with open(f) as fh:
            txt = fh.read().strip()
        obj = json.loads(txt if txt.startswith("{") and "\n{" not in txt else txt.splitlines()[0])
        assert isinstance(obj, dict) and obj, f
    cfg4 = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench_cfg4_shards_1gpu.json")))
    assert cfg4, "no cfg4 gene-shard record of this round"
    with open(cfg4[-1]) as fh:
        d = json.loads(fh.readline())
    assert d["config"]["mode"] == "shards" and d["roofline"]["frac"] > 0 and "50000 genes x 500 samples" in d["config"]["workload"]
