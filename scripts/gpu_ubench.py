"""Timing of the update kernel with parts ablated (development aid; ablated builds only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
for chains in (1, 4):
    ms, t = m.bench_gene_kernel(chains, 40, 30, 301)
    print(os.environ.get("PPCX_LIB"), "chains", chains, "update kernel us/launch", 1e3 * ms, "cmd type", t)
