"""Development aid: cfg3 fits (8 chains, 150 + 250; chain ids from OFFSET: rank r of a multi-GPU bench runs OFFSET = 8 r) over seeds -- wall time, gradient evaluations, rounds, the chains' step
sizes after warmup and tree depths: how often does a fit end warmup with a chain whose step size is far below the others'?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
if os.environ.get("LANES"):                    # lanes per gene of the log-likelihood launch (default: the library's choice)
    m.set_launch(int(os.environ["LANES"]), 0)
for seed in [int(x) for x in os.environ.get("SEEDS", "1,2,3,4,5,6,7,8,9,10,11,12").split(",")]:
    t0 = time.perf_counter()
    f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=seed, chain_id_offset=int(os.environ.get("OFFSET", 0)))
    dt = time.perf_counter() - t0
    dg = f.diagnostics(); kt = f.kernel_times()
    eps = dg["stepsize"][:, -1]; td = dg["treedepth"][:, 150:]
    print(f"seed {seed}: wall {dt:.2f} s, grad evals {f.timing().grad_evals}, rounds {kt['launch_triples']}, step sizes {np.array2string(eps, precision=4)}, "
          f"mean depth {td.mean():.2f} max {td.max()}, div {int(dg['divergent'][:, 150:].sum())}", flush=True)
    f.close()
m.close()
