"""Timing of the step / update kernels (development aid). With a NOSTEP build the state machine is skipped."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
for chains in (1, 4):
    for which, name in [(3, "step"), (4, "update"), (1, "close")]:
        ms, t = m.bench_gene_kernel(chains, 40, 30, 100 * which + 1)
        print(os.environ.get("PPCX_LIB", "default"), "chains", chains, name, "us/launch %.2f" % (1e3 * ms), "cmd type", t)
