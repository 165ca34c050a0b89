"""Development aid: one cfg3 fit (8 chains) at a given seed / lanes per gene / round structure: per-chain step sizes over warmup,
tree depths, divergences -- to see where a chain's adaptation goes wrong."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
if os.environ.get("LANES"):
    m.set_launch(int(os.environ["LANES"]), 0)
f = m.fit_nuts(chains=8, iter=int(os.environ.get("ITER", 400)), warmup=150, seed=int(os.environ.get("SEED", 2)))
dg = f.diagnostics()
np.set_printoptions(linewidth=250, precision=4, suppress=True)
print("grad evals", f.timing().grad_evals, "seconds", round(f.timing().seconds, 2))
for it in (0, 5, 10, 20, 40, 74, 75, 76, 90, 99, 100, 101, 120, 148, 149, 150, 151, 200):
    if it < dg["stepsize"].shape[1]:
        print(f"iter {it:3d}: stepsize {dg['stepsize'][:, it]}  depth {dg['treedepth'][:, it]}  accept {dg['accept'][:, it]}")
print("leapfrogs per chain", dg["n_leapfrog"].sum(1), "divergent (all iterations)", dg["divergent"].sum(1))
print("lp at the end", dg["lp"][:, -1])
