"""Development aid: the log-likelihood launch of `~ group + age` at BASELINE size (scripts/gpu_continuous_time.py's model) by the launch
plan's weight of a pass of genes with slopes (testing build: slope_cost_permille), kernel-level timing at warmed-up positions."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from ppcseq_amd import _lib as L, build
from ppcseq_amd.synth import synth
L.use_library(os.environ.get("PPCX_LIB") or build.build_testing())
G, S, K = 20000, 200, 1000
d2 = synth(G, S, seed=20253)
rng = np.random.default_rng(5)
age = rng.normal(0, 1, S); age = (age - age.mean()) / age.std()
X3 = np.concatenate([d2["X"], age[:, None]], axis=1)
for k, w in [(int(a), int(b)) for a, b in (x.split(":") for x in os.environ.get("KW", "1000:2500,1000:3500,1000:5000,1000:8000,0:2500,20000:2500").split(","))]:
    L.testing_set("slope_cost_permille", w)
    m = L.Model(d2["counts"], X3, d2["exposure"], k)
    ms = min(m.bench_kernel(0, 8, 3000, 60, 1)[0] for r in range(3))
    lanes, nb, b = m.get_plan(8)
    npass = np.diff(b) / (64 // lanes)
    print("K", k, "weight", w / 1000, "loglik us/launch (8 chains)", round(1e3 * ms, 2), "passes of the first 8 wavefronts", npass[:8].tolist(), "of the last", npass[-3:].tolist(), flush=True)
    m.close()
