"""A design beyond two groups at BASELINE size: 20 000 genes x 200 samples, `~ a` with a three-level factor (C = 3; model.matrix,
R/utilities.R:887-900), K = 1000 checked genes. Times the log-likelihood launch (8 chains, kernel-level) and whole fits
(8 chains, 150 + 250) on (a) the per-cell-exp path with the three-launch round -- what every C >= 3 design ran before round 4,
forced here through the testing build -- and (b) the factorised indicator path with pipelined rounds (the product's choice),
and the two-group design of cfg3 beside them. Writes gpurun_out/factor_design_c3.json (kept as profiles/rNN_factor_design_c3.json) (run on the GPU box)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppcseq_amd import _lib as L, build
from oracle import independent as ind
from ppcseq_amd.synth import synth

L.use_library(build.build_testing())
G, S, K = 20000, 200, 1000
d3 = ind.synth_factor(G, S, K, (3,), 20253)
d2 = synth(G, S, seed=20253)
out = {"workload": f"{G} genes x {S} samples, K = {K}, 8 chains; C = 3: three-level factor; C = 2: cfg3's two groups"}
for tag, d, force in (("c3_generic_three_launch", d3, 1), ("c3_indicator_pipelined", d3, 0), ("c2_two_group_pipelined", d2, 0)):
    L.testing_set("force_generic", force)
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    ms = min(m.bench_kernel(0, 8, 40 if r == 0 else 10, 100, 1)[0] for r in range(4))
    fits = []
    for seed in (1, 2):
        t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=seed); dt = time.perf_counter() - t0
        tm = f.timing(); f.close()
        fits.append({"seconds": round(dt, 3), "grad_evals": tm.grad_evals, "us_per_grad_eval_per_chain": round(1e6 * dt * 8 / tm.grad_evals, 2)})
    out[tag] = {"pipelined": m.get_rounds(8)[0], "loglik_launch_us_8_chains": round(1e3 * ms, 2), "fits": fits}
    print(tag, out[tag], flush=True)
    m.close()
L.testing_set("force_generic", 0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "factor_design_c3.json"), "w"), indent=1)     # merged back by gpurun; copied to profiles/rNN_factor_design_c3.json
