"""Development aid: where the device's ADVI run and the oracle's differ after convergence (tests/test_gpu_parity.py::test_advi_follows_oracle)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from oracle import independent as ind
from oracle.oracle import Oracle
O = Oracle()
d = ind.synth(40, 10, K=4, seed=21)
mo = O.model(d["counts"], d["X"], d["exposure"], 4, n_threads=4)
m = L.Model(d["counts"], d["X"], d["exposure"], 4)
for it in (50000, 400, 1000, 2000):
    ro = O.advi(mo, output_samples=400, seed=3, iter=it)
    f = m.fit_advi(output_samples=400, seed=3, iter=it)
    dr = f.draws()[0]
    diff = np.abs(dr - ro["draws"]); sd = ro["draws"].std(0)
    rel = diff / (1 + np.abs(ro["draws"]))
    worst = np.argsort(-rel.max(0))[:6]
    print("iter", it, "iterations", f.advi_info()["iterations"], ro["iterations"], "max rel", rel.max(), "max diff/sd", (diff / sd).max())
    for c in worst:
        print("   col", c, "of", dr.shape[1], "rel", rel[:, c].max(), "diff/sd", (diff[:, c] / sd[c]).max(), "sd", sd[c], "mean", ro["draws"][:, c].mean())
    f.close()
