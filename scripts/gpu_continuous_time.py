"""A continuous covariate at BASELINE size: 20 000 genes x 200 samples, `~ group + age` (C = 3: cfg3's two groups plus a
standardised numeric column; model.matrix, R/utilities.R:887-900), K = 1000 checked genes -- whose cells form eta per cell, an
exp each. Times the log-likelihood launch (8 chains, kernel level, positions of a warmed-up run) and whole fits (8 chains,
150 + 250) for (a) the three-launch round, what such a model ran through round 4, (b) pipelined rounds (round 5: the
coefficients kept among the coordinates' constants), and the two-group design of cfg3 beside them.
Writes profiles/r05_continuous_design.json (run on the GPU box)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppcseq_amd import _lib as L, build
from ppcseq_amd.synth import synth

L.use_library(build.build_testing())
G, S, K = 20000, 200, 1000
d2 = synth(G, S, seed=20253)
rng = np.random.default_rng(5)
age = rng.normal(0, 1, S); age = (age - age.mean()) / age.std()
X3 = np.concatenate([d2["X"], age[:, None]], axis=1)
out = {"workload": f"{G} genes x {S} samples, K = {K}, 8 chains; `~ group + age`: cfg3's counts and groups plus a standardised numeric column"}
for tag, X, pipe in (("group_age_three_launch", X3, 0), ("group_age_pipelined", X3, -1), ("two_group_pipelined", d2["X"], -1)):
    m = L.Model(d2["counts"], X, d2["exposure"], K)
    m.set_rounds(pipelined=pipe)
    ms = min(m.bench_kernel(0, 8, 3000, 60, 1)[0] for r in range(3))
    fits = []
    for seed in (1, 3):
        t0 = time.perf_counter(); f = m.fit_nuts(chains=8, iter=400, warmup=150, seed=seed); dt = time.perf_counter() - t0
        tm = f.timing(); dg = f.diagnostics(); f.close()
        fits.append({"seconds": round(dt, 3), "grad_evals": tm.grad_evals, "us_per_grad_eval_per_chain": round(1e6 * dt * 8 / tm.grad_evals, 2),
                     "divergent_after_warmup": int(dg["divergent"][:, 150:].sum()), "max_treedepth": int(dg["treedepth"][:, 150:].max())})
    out[tag] = {"pipelined": bool(m.get_rounds(8)[0]), "loglik_launch_us_8_chains": round(1e3 * ms, 2), "fits": fits}
    print(tag, out[tag], flush=True)
    m.close()
a, b = out["group_age_pipelined"]["loglik_launch_us_8_chains"], out["two_group_pipelined"]["loglik_launch_us_8_chains"]
out["launch_ratio_group_age_over_two_group"] = round(a / b, 3)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_continuous_design.json"), "w"), indent=1)   # merged back by gpurun; copied to profiles/
print("ratio", out["launch_ratio_group_age_over_two_group"])
