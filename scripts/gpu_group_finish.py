"""Development aid: when do the chain groups of a cfg3 fit finish (progress callback)? A group that the hardware serves first
leaves the last stretch of the fit to fewer groups."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=1000); f.close()
for seed in [int(x) for x in os.environ.get("SEEDS", "1,2,3").split(",")]:
    last = {}
    def cb(c0, n, done, rounds, sec):
        last[c0] = (n, done, rounds, round(sec, 3))
    m.set_progress(cb, every_seconds=0.05)
    t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=seed); dt = time.perf_counter() - t0
    leap = f.diagnostics()["n_leapfrog"].sum(axis=1)
    f.close()
    print(f"seed {seed}: wall {dt:.3f} s; groups (first chain: chains, done, rounds, seconds at the last report): {last}; leapfrogs per chain {leap.tolist()}", flush=True)
