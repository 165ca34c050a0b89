"""Why the README case (README.md:50-92) carries borderline calls beyond CYP1A1 and LYZ: for MMP8 and CCNA1 (the genes that are
called in some runs) and for the two README genes, per sampler seed and per mode -- the README's own
(ADVI + approximated analysis) and NUTS with the full posterior -- the cell that sticks out furthest above its interval in the test
pass: its count and the interval's upper end, on the GPU (identify_outliers) and on the oracle (tests/test_oracle_reference_cases.py's
oracle-driven procedure; fewer seeds: CPU). Writes gpurun_out/readme_margins.json (copied to profiles/r05_readme_margins.json)."""
import json, os, sys
import numpy as np
import pandas as pd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppcseq_amd.methods import identify_outliers

WATCH = ("CYP1A1", "LYZ", "MMP8", "CCNA1")
z = np.load(os.path.join(ROOT, "tests", "golden", "counts_bundled.npz"), allow_pickle=False)
b = {k: z[k] for k in z.files}
genes, samples = [str(g) for g in b["genes"]], [str(s) for s in b["samples"]]
G, S = len(genes), len(samples)
df = pd.DataFrame({"symbol": np.repeat(genes, S), "sample": np.tile(samples, G), "value": b["value"].reshape(-1),
                   "PValue": np.repeat(b["PValue"], S), "FDR": np.repeat(b["FDR"], S), "Label": np.tile(b["Label"].astype(str), G)})
df["is_significant"] = df["FDR"] < 0.01
n_gpu, n_cpu = int(os.environ.get("SEEDS", 8)), int(os.environ.get("ORACLE_SEEDS", 2))
MODES = (("readme_defaults_advi_approximated", {}, dict(vb=True, approx_analysis=True)),
         ("nuts_full_posterior", dict(approximate_posterior_inference=False, approximate_posterior_analysis=False), dict(vb=False, approx_analysis=False)))
out = {"note": "per gene: the cell furthest above its interval in the test pass (count / .upper) and the cell furthest below it (.lower / count); a ratio "
               "above 1 is outside the interval, and the cell is called when that is the direction of the gene's slope"}


def worst(counts_row, lower_row, upper_row, called_row):
    """the cell furthest above its interval and the cell furthest below it (ratio > 1: outside), and the cells called"""
    hi = counts_row / np.maximum(upper_row, 1e-9)
    lo = lower_row / np.maximum(counts_row, 0.5)
    a, b = int(np.argmax(hi)), int(np.argmax(lo))
    return {"above": {"sample": samples[a], "count": int(counts_row[a]), "upper": round(float(upper_row[a]), 1), "count_over_upper": round(float(hi[a]), 3)},
            "below": {"sample": samples[b], "count": int(counts_row[b]), "lower": round(float(lower_row[b]), 1), "lower_over_count": round(float(lo[b]), 3)},
            "called_samples": [samples[i] for i in np.flatnonzero(called_row)]}


for mode, kw, okw in MODES:
    runs = []
    for seed in range(1, n_gpu + 1):
        res = identify_outliers(df, formula="~ Label", sample="sample", transcript="symbol", abundance="value", significance="PValue",
                                do_check="is_significant", percent_false_positive_genes=5, cores=4, seed=seed, **kw)
        by = res.set_index("symbol")
        rec = {"seed": seed}
        for g in WATCH:
            sw = by.loc[g, "sample_wise_data"]
            rec[g] = worst(sw["value"].to_numpy(float), sw[".lower"].to_numpy(float), sw[".upper"].to_numpy(float), sw["deleterious_outliers"].to_numpy(bool))
        runs.append(rec)
        print("gpu", mode, rec, flush=True)
    out.setdefault("gpu", {})[mode] = runs
if n_cpu > 0:
    from oracle.oracle import Oracle
    O = Oracle()
    from tests.test_oracle_reference_cases import _oracle_identify_outliers, _readme_selection
    counts, X, names = _readme_selection(b)
    for mode, kw, okw in MODES:
        runs = []
        for seed in range(1, n_cpu + 1):
            r1, r2 = _oracle_identify_outliers(O, counts, X, 15, pfp=5, cores=4, seed=seed, **okw)
            rec = {"seed": seed}
            for g in WATCH:
                i = names.index(g)
                rec[g] = worst(counts[i].astype(float), np.asarray(r2.lower[i], float), np.asarray(r2.upper[i], float), np.asarray(r2.deleterious_outliers[i], bool))
            runs.append(rec)
            print("oracle", mode, rec, flush=True)
        out.setdefault("oracle", {})[mode] = runs
summary = {}
for side in ("gpu", "oracle"):
    for mode, runs in out.get(side, {}).items():
        summary[f"{side}/{mode}"] = {g: {"count_over_upper_min_max": [min(r[g]["above"]["count_over_upper"] for r in runs), max(r[g]["above"]["count_over_upper"] for r in runs)],
                                         "lower_over_count_min_max": [min(r[g]["below"]["lower_over_count"] for r in runs), max(r[g]["below"]["lower_over_count"] for r in runs)],
                                         "called_in": sum(len(r[g]["called_samples"]) > 0 for r in runs), "runs": len(runs)} for g in WATCH}
out["summary"] = summary
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "readme_margins.json"), "w"), indent=1)
print(json.dumps(out["summary"], indent=1))
