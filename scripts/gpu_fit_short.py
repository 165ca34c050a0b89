"""Development aid: one short cfg3 fit (mode from PPCX_PIPELINE / PPCX_STREAM_GROUPS) for profiling."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
d = synth(G, S, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
if os.environ.get("LANES"):
    m.set_launch(int(os.environ["LANES"]), 0)
f = m.fit_nuts(chains=int(os.environ.get("CHAINS", 8)), iter=int(os.environ.get("ITER", 60)), warmup=int(os.environ.get("WARMUP", 30)), seed=int(os.environ.get("SEED", 1)))
kt, tm = f.kernel_times(), f.timing()
print("plan", m.get_plan(int(os.environ.get("CHAINS", 8)))[:2], "pipeline", os.environ.get("PPCX_PIPELINE", "default"), "rounds", kt["launch_triples"], "pump s", round(tm.seconds, 3),
      "us/round", round(1e6 * tm.seconds / kt["launch_triples"], 1), "grad evals", tm.grad_evals)
