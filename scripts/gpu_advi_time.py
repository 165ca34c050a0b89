"""ADVI (the reference's default inference mode, rstan::vb) at cfg3 size: time to convergence (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
d = synth(G, S, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
t0 = time.perf_counter(); f = m.fit_advi(output_samples=1000, seed=3); t1 = time.perf_counter()
info = f.advi_info()
dr = f.draws()[0]
tr = d["truth"]
ci = f.ppc(1.0, 0.05, 0.95, seed=4)
t2 = time.perf_counter()
print(json.dumps({"config": f"ADVI mean-field, synthetic {G} x {S} (seed 20253), 1000 output samples", "seconds_fit": round(t1 - t0, 2),
                  "seconds_ppc": round(t2 - t1, 2), "iterations": int(info["iterations"]), "converged": bool(info["converged"]),
                  "eta": float(info["eta"]), "elbo": float(info["elbo"]),
                  "corr_intercept_mean_vs_truth": round(float(np.corrcoef(dr[:, 3:3 + G].mean(0), tr["intercept"])[0, 1]), 4),
                  "corr_sigma_raw_mean_vs_truth": round(float(np.corrcoef(dr[:, -3 - G:-3].mean(0), tr["sigma_raw"])[0, 1]), 4)}))
