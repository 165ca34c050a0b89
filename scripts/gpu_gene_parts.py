"""Development aid: the gene kernel's parts at kernel level (testing build): whole, without the anticipated constants, the command's
coordinate work only. Under scripts/gpu_sq_pmc.sh the dispatches' SQ_INSTS_VALU give the parts' instruction counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L, build
from ppcseq_amd.synth import synth
L.use_library(os.environ.get("PPCX_LIB") or build.build_testing())
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
chains = int(os.environ.get("CHAINS", 8))
parts = (("whole, no level", 8, 0), ("whole, 1 level", 8, 1), ("whole, 3 levels", 8, 3), ("no anticipated constants", 10, 0), ("coordinate work only", 11, 0), ("without proposal copies", 9, 0), ("first leaf of a transition (fresh momenta)", 12, 0))
sel = [int(x) for x in os.environ["PARTS"].split(",")] if os.environ.get("PARTS") else range(len(parts))
for tag, which, nm in [parts[i] for i in sel]:
    ms = min(m.bench_kernel(which, chains, 40 if r == 0 else 10, int(os.environ.get("REPS", 200)), nm)[0] for r in range(3))
    print(f"chains {chains} {tag}: {1e3 * ms:.2f} us", flush=True)
