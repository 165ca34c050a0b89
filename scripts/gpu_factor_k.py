"""Development aid: what a pass of genes with slopes costs in a factor design (C = 3), and how the launch plan's weight for it
moves the log-likelihood launch (kernel-level timing, testing build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from ppcseq_amd import _lib as L, build
from oracle import independent as ind
from ppcseq_amd.synth import synth
L.use_library(build.build_testing())
G, S = 20000, 200
d3 = ind.synth_factor(G, S, 1000, (3,), 20253)
d2 = synth(G, S, seed=20253)
for tag, d, K, w in (("C2 K0", d2, 0, 0), ("C2 K1000", d2, 1000, 0), ("C3 K0", d3, 0, 0), ("C3 K20000", d3, 20000, 0),
                     ("C3 K1000 w1.0", d3, 1000, 1000), ("C3 K1000 w1.45", d3, 1000, 1450), ("C3 K1000 w2.0", d3, 1000, 2000), ("C3 K1000 w3.0", d3, 1000, 3000), ("C3 K1000 w5.0", d3, 1000, 5000)):
    L.testing_set("slope_cost_permille", w)
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    ms = min(m.bench_kernel(0, 8, 40 if r == 0 else 10, 100, 1)[0] for r in range(4))
    lanes, nb, b = m.get_plan(8)
    npass = np.diff(b) / (64 // lanes)
    print(tag, "loglik us/launch (8 chains)", round(1e3 * ms, 2), "passes of the first 8 wavefronts", npass[:8].tolist(), "of the last", npass[-3:].tolist(), flush=True)
    m.close()
