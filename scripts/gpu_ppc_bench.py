"""The posterior-predictive kernel at the bench's workload (cfg3: K = 1000 checked genes x 200 samples, 8 chains x 250 kept
draws = 2000 predictive draws per cell) and at a longer one (n_gen via N_GEN, resampled): kernel ms (HIP events) and draws/s."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=1)
out = {}
for tag, kw in (("full_2000", {}), ("approx_resampled_10500", dict(n_gen=10500, resample=True)), ("approx_resampled_4000", dict(n_gen=4000, resample=True))):
    ms = []
    for rep in range(3):
        f.ppc(0.7352941, 0.0025, 0.9975, seed=3, **kw)
        k, n = f.ppc_timing(); ms.append(k)
    out[tag] = {"kernel_ms": round(min(ms), 3), "nb_draws": n, "G_draws_per_s": round(n / min(ms) / 1e6, 2)}
    print(tag, out[tag], flush=True)
print(json.dumps(out))
