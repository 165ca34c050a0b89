"""Generator of tests/golden/cfg3_nuts_oracle.npz: the ORACLE's NUTS (oracle/ppc_oracle.c, Stan defaults, max_treedepth 10) on
BASELINE config 3 -- synthetic 20 000 genes x 200 samples, seed 20253 -- for the first 45 warm-up iterations of 2 chains at
seed 20253. One oracle gradient costs ~0.25 s on 8 host cores and the run needs several thousand of them, so the result is
committed as a fixture instead of being recomputed by the GPU test (tests/test_gpu_configs.py), which compares tree sizes,
tree depths, divergences and step sizes iteration by iteration.

    python tests/golden/make_cfg3_nuts_fixture.py [threads]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle            # noqa: E402
from ppcseq_amd.synth import synth          # noqa: E402

G, S, DATA_SEED, CHAINS, ITER, SEED = 20000, 200, 20253, 2, 45, 20253
threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
d = synth(G, S, seed=DATA_SEED)
O = Oracle(native=True)
m = O.model(d["counts"], d["X"], d["exposure"], d["K"], n_threads=threads)
t0 = time.time()
r = O.nuts_model(m, O.cfg(chains=CHAINS, iter=ITER, warmup=ITER, seed=SEED))
print("oracle NUTS: %d gradient evaluations in %.0f s" % (int(r.n_leapfrog.sum()), time.time() - t0))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cfg3_nuts_oracle.npz"),
                    n_leapfrog=r.n_leapfrog, treedepth=r.treedepth, divergent=r.divergent, stepsize=r.stepsize, accept=r.accept,
                    config=np.array([G, S, DATA_SEED, CHAINS, ITER, SEED]))
print(r.n_leapfrog.tolist())
