"""Development aid: how unevenly the chains of a cfg3 fit finish (leapfrogs per chain) and what that costs: the rounds a fit needs
are those of its slowest chain, and a round with few chains still running costs much more per chain."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
cost = {1: 28.5, 2: 35.3, 3: 43.2, 4: 50.0, 5: 59.4, 6: 62.9, 7: 68.0, 8: 72.6}     # us per single-stream round by chains active
for groups in (1, 0):
    m.set_rounds(stream_groups=groups)
    for seed in (1, 2, 3):
        t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=seed); dt = time.perf_counter() - t0
        nl = f.diagnostics()["n_leapfrog"].sum(axis=1); f.close()
        srt = np.sort(nl)
        est = sum((srt[i] - (srt[i - 1] if i else 0)) * cost[8 - i] for i in range(8)) * 1e-6
        print(f"groups {groups or 'default'} seed {seed}: wall {dt:.3f} s; leapfrogs per chain min {nl.min()} mean {nl.mean():.0f} max {nl.max()}; "
              f"single-stream model: {est:.3f} s, of which {est - nl.mean() * 72.6e-6:.3f} s above mean x 72.6 us; sorted {srt.tolist()}", flush=True)
