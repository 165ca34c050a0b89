import sys, os
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd.methods import identify_outliers
z = np.load("tests/golden/counts_bundled.npz")
genes = [str(g) for g in z["genes"]]; samples = [str(s) for s in z["samples"]]
G, S = len(genes), len(samples)
df = pd.DataFrame({"symbol": np.repeat(genes, S), "sample": np.tile(samples, G), "value": z["value"].reshape(-1),
                   "PValue": np.repeat(z["PValue"], S), "FDR": np.repeat(z["FDR"], S), "Label": np.tile(z["Label"].astype(str), G)})
df["is_significant"] = df["FDR"] < 0.01
for seed in [7, 8, 9]:
    res = identify_outliers(df, formula="~ Label", sample="sample", transcript="symbol", abundance="value",
                            significance="PValue", do_check="is_significant", percent_false_positive_genes=5, cores=4, seed=seed)
    print("seed", seed, res[["symbol", "ppc_samples_failed", "tot_deleterious_outliers"]].query("ppc_samples_failed>0").values.tolist())
    for g in ["MMP8", "CYP1A1", "LYZ"]:
        sw = res.set_index("symbol").loc[g, "sample_wise_data"]
        bad = sw[~sw["posterior_predictive_check_succeded"]]
        print("  ", g, bad[["sample", "value", ".lower", ".upper", "deleterious_outliers", "Label"]].values.tolist(), "slope", sw["slope_after_outlier_filtering"].iloc[0])
    if seed == 7:
        sw = res.set_index("symbol").loc["MMP8", "sample_wise_data"]
        print(sw[["sample", "value", ".lower", ".upper", "Label", "exposure_rate"]].to_string())
