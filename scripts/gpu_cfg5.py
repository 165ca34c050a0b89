"""BASELINE cfg5 end to end on the GPU: the cfg3 matrix (20 000 x 200, seed 20253) through identify_outliers() --
discovery pass, outlier exclusion, test pass with truncation compensation -- with percent_false_positive_genes = 5 and
.do_check = the first K genes. The CPU path cannot run this size (hours per fit), so the check is against the generator's
truth: the injected outliers must come back as deleterious outliers and the clean checked genes must not.
Prints one JSON line (kept under profiles/)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandas as pd
from ppcseq_amd.synth import synth
from ppcseq_amd.methods import identify_outliers

G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
d = synth(G, S, seed=20253)
K = d["K"]
t0 = time.perf_counter()
genes = np.array([f"g{i:05d}" for i in range(G)])
samples = np.array([f"s{j:03d}" for j in range(S)])
gi, sj = np.meshgrid(np.arange(G), np.arange(S), indexing="ij")
rng = np.random.default_rng(1)
pval = np.concatenate([np.full(K, 1e-6), rng.uniform(0.01, 1, G - K)])
df = pd.DataFrame({"symbol": genes[gi.ravel()], "sample": samples[sj.ravel()], "value": d["counts"].ravel().astype(np.int64),
                   "Label": np.where(d["X"][sj.ravel(), 1] > 0.5, "B", "A"), "PValue": pval[gi.ravel()],
                   "is_significant": (gi.ravel() < K)})
t_frame = time.perf_counter() - t0
t0 = time.perf_counter()
res = identify_outliers(df, formula="~ Label", sample="sample", transcript="symbol", abundance="value", significance="PValue",
                        do_check="is_significant", percent_false_positive_genes=5, how_many_negative_controls=G - K,
                        seed=int(os.environ.get("SEED", 20255)), cores=int(os.environ.get("CORES", 8)),
                        approximate_posterior_inference=False, approximate_posterior_analysis=False)
t_all = time.perf_counter() - t0
flag = {r["symbol"]: r["tot_deleterious_outliers"] for _, r in res.iterrows()}
inj_genes = sorted({g for g, _ in d["injected"]})
hit = 0
for g, s in d["injected"]:
    sw = res.loc[res["symbol"] == genes[g], "sample_wise_data"].iloc[0]
    hit += bool(sw["deleterious_outliers"].to_numpy()[s])
clean = [i for i in range(K) if i not in set(inj_genes)]
fp_genes = sum(flag[genes[i]] > 0 for i in clean)
dg1, dg2 = res.attrs["diagnostics_discovery"], res.attrs["diagnostics_test"]
def leap(dg):
    return int(np.asarray(dg["n_leapfrog"]).sum()) if isinstance(dg, dict) and "n_leapfrog" in dg else None
print(json.dumps({
    "config": f"cfg5: synthetic {G} x {S} (seed 20253), K = {K} checked genes, ~ Label, percent_false_positive_genes = 5, two passes",
    "seconds_total": round(t_all, 2), "seconds_building_the_tidy_frame": round(t_frame, 2),
    "total_draws_test_pass": int(res.attrs["total_draws"]),
    "injected_outlier_cells": len(d["injected"]), "injected_cells_flagged_deleterious": int(hit),
    "clean_checked_genes": len(clean), "clean_checked_genes_with_a_deleterious_outlier": int(fp_genes),
    "false_positive_gene_rate": round(fp_genes / max(len(clean), 1), 4),
    "sampler_seed": int(os.environ.get("SEED", 20255)),
    "clean_checked_genes_flagged": [int(i) for i in clean if flag[genes[i]] > 0],
    "leapfrogs_discovery": leap(dg1), "leapfrogs_test": leap(dg2)}))
