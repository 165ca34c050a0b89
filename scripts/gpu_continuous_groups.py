"""Development aid: whole fits of `~ group + age` at BASELINE size (scripts/gpu_continuous_time.py's model) by number of chain groups and
lanes per gene."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
G, S, K = 20000, 200, 1000
d2 = synth(G, S, seed=20253)
rng = np.random.default_rng(5)
age = rng.normal(0, 1, S); age = (age - age.mean()) / age.std()
X3 = np.concatenate([d2["X"], age[:, None]], axis=1)
m = L.Model(d2["counts"], X3, d2["exposure"], K)
m.fit_nuts(chains=8, iter=30, warmup=20, seed=9).close()
for groups in (1, 2, 3):
    for lanes in (0, 4, 8):
        m.set_rounds(stream_groups=groups); m.set_launch(lanes, 0)
        t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=1); dt = time.perf_counter() - t0
        ge = f.timing().grad_evals; f.close()
        print(f"groups {groups} lanes {lanes} (plan {m.get_launch()}): fit {dt:.3f} s, {1e6 * dt * 8 / ge:.2f} us per round of 8", flush=True)
