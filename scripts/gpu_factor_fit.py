"""Development aid: whole fits (8 chains, 150 + 250) of the three-level factor design at BASELINE size with the build PPCX_LIB
names (default: the product) -- the four-column instantiations of the kernels. Seeds from SEEDS (1,2)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from oracle import independent as ind
d = ind.synth_factor(20000, 200, 1000, (3,), 20253)
m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
for seed in [int(x) for x in os.environ.get("SEEDS", "1,2").split(",")]:
    t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=seed); dt = time.perf_counter() - t0
    print(f"{os.path.basename(L.LIB_PATH)} seed {seed}: {dt:.3f} s, grad evals {f.timing().grad_evals}, lp sha {hashlib.sha1(f.diagnostics()['lp'].tobytes()).hexdigest()[:10]}", flush=True)
    f.close()
m.close()
