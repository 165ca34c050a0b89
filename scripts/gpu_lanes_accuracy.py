"""Development aid: log density and gradient at cfg3 size by lanes per gene (4, 8, 16, 32) against the optimised CPU comparator
(test infrastructure) at points of a realistic spread -- does any lane count stand out in accuracy?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
from oracle import oracle as O
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
cf = O.CpuFast(); mo = cf.model(d["counts"], d["X"], d["exposure"], d["K"])
rng = np.random.default_rng(5)
f = m.fit_nuts(chains=2, iter=60, warmup=50, seed=3)          # points of the typical set, not U(-2, 2)
pts = f.draws()[:, -3:, :].reshape(-1, m.D)
f.close()
pts = np.vstack([pts, rng.uniform(-2, 2, (2, m.D))])
ref = [cf.log_prob_grad(mo, p, threads=16) for p in pts]
for lanes in (4, 8, 16, 32):
    m.set_launch(lanes, 0)
    lp, g = m.log_prob_grad(pts)
    e_lp = max(abs(lp[i] - ref[i][0]) / abs(ref[i][0]) for i in range(len(pts)))
    e_g = max(np.max(np.abs(g[i] - ref[i][1]) / (np.abs(ref[i][1]) + 1e-6 * np.max(np.abs(ref[i][1])))) for i in range(len(pts)))
    e_abs = max(np.max(np.abs(g[i] - ref[i][1])) for i in range(len(pts)))
    print(f"lanes per gene {lanes}: lp rel {e_lp:.2e}, grad rel max {e_g:.2e}, grad abs max {e_abs:.2e}", flush=True)
