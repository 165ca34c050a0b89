"""Kernel-level timing of the loglik kernel over launch geometries (development aid)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
G, S = int(os.environ.get("G", 20000)), int(os.environ.get("S", 200))
d = synth(G, S, seed=20253); K = d["K"]
if os.environ.get("NO_SLOPES"): K = 0                     # every gene takes the intercept-only cell path
d["counts"] //= int(os.environ.get("COUNT_DIV", 1))      # > 1: low counts everywhere (times the small-count regime)
m = L.Model(d["counts"], d["X"], d["exposure"], K)
bgrad = 4.0 * G * S + 16.0 * 3 * G + 8.0 * S * 3
for chains in [int(x) for x in os.environ.get("CHAINS", "4").split(",")]:
    for lanes in [int(x) for x in os.environ.get("LANES", "8,16").split(",")]:
        for gpw in [int(x) for x in os.environ.get("GPW", "0").split(",")]:
            m.set_launch(lanes, gpw)
            ms, t = m.bench_gene_kernel(chains, 40, 30, 1)
            print(f"chains {chains} L {lanes} gpw {gpw} launch {m.get_launch()} type {t}: loglik {1e3*ms/chains:.2f} us/chain-grad ; {bgrad*chains/ms/1e6:.0f} GB/s", flush=True)
