"""Per-kernel timings inside a short real fit (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
for chains in (4,):
    f = m.fit_nuts(chains=chains, iter=60, warmup=60, seed=1)
    kt = f.kernel_times(); tm = f.timing()
    print("chains", chains, {k: round(1e3 * v, 2) if k != "launch_triples" else v for k, v in kt.items()}, "wall", round(tm.seconds, 3), "us/triple", round(1e6 * tm.seconds / kt["launch_triples"], 1))
