"""Kernel-level timing of the loglik kernel over launch geometries (development aid). Configurations are timed in
interleaved rounds (ROUNDS) and the minimum and median per configuration are printed: single timings on a shared box move
by several per cent with the clock."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd import build as _b
L.use_library(os.environ.get("PPCX_LIB") or _b.build_testing())          # kernel-level timing lives in the testing build
from ppcseq_amd.synth import synth
G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
d = synth(G, S, seed=20253); K = d["K"]
if os.environ.get("NO_SLOPES"): K = 0                     # every gene takes the intercept-only cell path
d["counts"] //= int(os.environ.get("COUNT_DIV", 1))      # > 1: low counts everywhere (times the small-count regime)
m = L.Model(d["counts"], d["X"], d["exposure"], K)
bgrad = 4.0 * G * S + 16.0 * 3 * G + 8.0 * S * 3
cfgs = [(c, l, w) for c in [int(x) for x in os.environ.get("CHAINS", "8").split(",")]
        for l in [int(x) for x in os.environ.get("LANES", "8").split(",")]
        for w in [int(x) for x in os.environ.get("WGS", "0").split(",")]]
res = {c: [] for c in cfgs}
reps = int(os.environ.get("REPS", 100))
for r in range(int(os.environ.get("ROUNDS", 5))):
    for cfg in cfgs:
        chains, lanes, wgs = cfg
        m.set_launch(lanes, wgs)
        ms, t = m.bench_kernel(0, chains, int(os.environ.get("WARM", 10 if r else 40)), reps, 1)
        res[cfg].append((ms, m.get_launch(), t))
for cfg in cfgs:
    chains, lanes, wgs = cfg
    ms = np.array([x[0] for x in res[cfg]])
    print(f"chains {chains} L {lanes} workgroups {wgs} launch {res[cfg][0][1]} type {res[cfg][0][2]}: loglik us/launch min {1e3*ms.min():.1f} median {1e3*np.median(ms):.1f} max {1e3*ms.max():.1f} ; "
          f"{1e3*ms.min()/chains:.2f} us/chain-grad ; {bgrad*chains/ms.min()/1e6:.0f} GB/s", flush=True)
