"""Kernel-level timing of every kernel of one leapfrog round (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd import build as _b
L.use_library(os.environ.get("PPCX_LIB") or _b.build_testing())          # kernel-level timing lives in the testing build
from ppcseq_amd.synth import synth
G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
chains = int(os.environ.get("CHAINS", 4))
d = synth(G, S, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
names = {0: "loglik", 1: "close", 2: "loglik+close", 3: "step", 4: "update", 5: "step:reduce", 6: "step:advance", 7: "step+update"}
for which, name in names.items():
    ms, t = m.bench_kernel(which, chains, 40, 100, 1)
    print(f"{name:14s} {1e3 * ms:8.2f} us/launch (chains {chains}, cmd type {t})", flush=True)
