"""Development aid: ONE chain of a cfg3 fit, iteration by iteration (step size, tree depth, leapfrogs, divergences, acceptance) --
a chain's draws do not depend on the chains it shares launches with, so chain CHAIN of the 8-chain fit at SEED is reproduced alone."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
m.set_launch(int(os.environ.get("LANES", 8)), 0)          # lanes per gene of the 8-chain fit
it = int(os.environ.get("ITER", 170))
f = m.fit_nuts(chains=1, iter=it, warmup=150, seed=int(os.environ.get("SEED", 2)), chain_id_offset=int(os.environ.get("CHAIN", 0)))
dg = f.diagnostics()
for i in range(it):
    print(f"{i:4d} eps {dg['stepsize'][0, i]:.5g} depth {dg['treedepth'][0, i]} leap {dg['n_leapfrog'][0, i]} div {dg['divergent'][0, i]} acc {dg['accept'][0, i]:.3f}")
dr = f.draws()[0]
G, K = 20000, d["K"]
print("hypers of the last draw", dr[-1, :3], dr[-1, -3:])
print("sd over the kept draws: intercept", np.percentile(dr[:, 3:3 + G].std(0), [0, 1, 50, 99, 100]), "sigma_raw", np.percentile(dr[:, 3 + G + K:3 + 2 * G + K].std(0), [0, 1, 50, 99, 100]))
