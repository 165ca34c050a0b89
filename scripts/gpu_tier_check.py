"""Development aid: log-likelihood kernel with and without the tail tiers (kernel time, lp/grad agreement), then fit times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd import build as _b
L.use_library(os.environ.get("PPCX_LIB") or _b.build_testing())          # kernel-level timing lives in the testing build
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
rng = np.random.default_rng(0)
res = {}
for tag, env in (("tiers", None), ("no tiers", "1")):
    L.testing_set("no_tail_tiers", 1 if env else 0)
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    u = np.random.default_rng(0).uniform(-0.3, 0.3, m.D); u[3:3 + 20000] += 6.0
    lp, g = m.log_prob_grad(u)
    ms = min(m.bench_kernel(0, 8, 40 if r == 0 else 10, 100, 1)[0] for r in range(5))
    res[tag] = (lp, g, ms)
    print(tag, "loglik us/launch (8 chains)", round(1e3 * ms, 2), "lp", lp, flush=True)
    for pipe in (1,):
        m.set_rounds(pipelined=-1 if pipe else 0)
        ts = []
        for rep in range(2):
            t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=1 + rep); ts.append(time.perf_counter() - t0)
            kt = f.kernel_times(); f.close()
        print(tag, "pipe", pipe, "fit s", [round(t, 3) for t in ts], "kernels us", {k: round(1e3 * v, 1) for k, v in kt.items() if k != "launch_triples"}, flush=True)
    m.close()
a, b = res["tiers"], res["no tiers"]
print("lp rel diff", abs(a[0] - b[0]) / abs(b[0]), "grad max rel diff", float(np.max(np.abs(a[1] - b[1]) / (1 + np.abs(b[1])))))
