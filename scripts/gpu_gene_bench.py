"""Development aid: the gene kernel of a pipelined round at kernel level (testing build), by chains per launch and levels the leaf
closes, with and without the proposal copies."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L, build
from ppcseq_amd.synth import synth
L.use_library(os.environ.get("PPCX_LIB") or build.build_testing())
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
for chains in [int(x) for x in os.environ.get("CHAINS", "8,1").split(",")]:
    for n_merge in (0, 1, 3):
        row = []
        for which in (8, 9):
            ms = [m.bench_kernel(which, chains, 40 if r == 0 else 10, 200, n_merge)[0] for r in range(4)]
            row.append(min(ms))
        print(f"chains {chains} levels closed {n_merge}: gene kernel {1e3 * row[0]:.2f} us, without proposal copies {1e3 * row[1]:.2f} us", flush=True)
