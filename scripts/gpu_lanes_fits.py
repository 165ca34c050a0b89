"""Development aid: whole cfg3 fits (150 + 250) by number of chains and lanes per gene -- seconds and microseconds per gradient
evaluation of a chain, the library's default chain groups. The lanes-per-gene rule of choose_launch (ppcx_capi.hip) is set from this."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(int(os.environ.get("G", 20000)), int(os.environ.get("S", 200)), seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
seeds = [int(x) for x in os.environ.get("SEEDS", "1,3").split(",")]
f = m.fit_nuts(chains=8, iter=40, warmup=20, seed=99); f.close()          # first-launch costs
for nch in [int(x) for x in os.environ.get("CHAINS", "1,2,3,4,8").split(",")]:
    for lanes in [int(x) for x in os.environ.get("LANES", "0,4,8,16").split(",")]:
        m.set_launch(lanes, 0)
        ts, us = [], []
        for seed in seeds:
            t0 = time.perf_counter()
            f = m.fit_nuts(chains=nch, iter=400, warmup=150, seed=seed)
            dt = time.perf_counter() - t0
            ge = f.timing().grad_evals
            stuck = bool((f.diagnostics()["treedepth"][:, 150:].mean(axis=1) > 9).any())
            f.close()
            if not stuck:
                ts.append(dt); us.append(1e6 * dt * nch / ge)
        print(f"chains {nch} lanes {lanes} (plan {m.get_launch()}): fit {np.mean(ts):.3f} s, {np.mean(us):.2f} us per gradient evaluation and chain ({len(ts)} fits)", flush=True)
