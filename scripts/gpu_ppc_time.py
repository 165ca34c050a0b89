"""Time split of a test pass at cfg5 size: NUTS fit vs posterior-predictive kernel (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
chains, per = 8, int(os.environ.get("DRAWS_PER_CHAIN", 2500))
t0 = time.perf_counter(); f = m.fit_nuts(chains=chains, iter=150 + per, warmup=150, seed=3); t1 = time.perf_counter()
print(f"fit: {t1 - t0:.2f} s for {chains} x {per} kept draws; {f.timing().grad_evals} gradient evaluations", flush=True)
for rep in range(2):
    t0 = time.perf_counter(); ci = f.ppc(0.7352941, 2.5e-4, 1 - 2.5e-4, seed=3); t1 = time.perf_counter()
    n = chains * per * d["K"] * 200
    print(f"ppc: {t1 - t0:.2f} s for {n:.3g} NB draws = {n / (t1 - t0) / 1e9:.2f} G draws/s", flush=True)
