"""First GPU contact: parity of lp/grad, NUTS and PPC against the oracle on small problems, then timings."""
import sys, time, json, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from oracle.oracle import Oracle
from oracle import independent as ind

out = {}
O = Oracle()
print("devices", L.device_count(), flush=True)
# ---- 1. lp/grad parity
for (G, S, C, K, seed) in [(7, 5, 2, 3, 1), (40, 21, 2, 5, 2), (30, 11, 3, 4, 3), (12, 6, 1, 2, 4), (300, 50, 2, 15, 6), (257, 200, 2, 13, 7)]:
    d = ind.synth(G, S, K=K, seed=seed, C=C)
    rng = np.random.default_rng(seed)
    D = O.dim(G, C, K)
    u = rng.uniform(-1, 1, (3, D)); u[:, 3:3 + G] += 5
    excl = np.array([1, 2 * S + 3, (G - 1) * S], dtype=np.int32) if seed % 2 == 0 else None
    mo = O.model(d['counts'], d['X'], d['exposure'], K, excl=excl)
    m = L.Model(d['counts'], d['X'], d['exposure'], K, excl=excl)
    for Lg in [0, 1, 4, 16, 64]:
        m.set_launch(Lg, 0)
        lp, g = m.log_prob_grad(u)
        for i in range(3):
            lpo, go = O.log_prob_grad(mo, u[i])
            rel = abs(lp[i] - lpo) / abs(lpo)
            ge = np.max(np.abs(g[i] - go) / (1 + np.abs(go)))
            assert rel < 1e-11 and ge < 1e-10, (G, S, C, K, Lg, rel, ge)
    print("lp/grad ok", G, S, C, K, m.get_launch(), flush=True)
    m.close()

# ---- 2. NUTS parity (same seed, same RNG spec): first iterations identical to rounding
G, S, C, K = 30, 8, 2, 4
d = ind.synth(G, S, K=K, seed=3, C=C)
mo = O.model(d['counts'], d['X'], d['exposure'], K)
cfg = O.cfg(chains=2, iter=200, warmup=150, seed=11)
r = O.nuts_model(mo, cfg)
m = L.Model(d['counts'], d['X'], d['exposure'], K)
t = time.time(); f = m.fit_nuts(chains=2, iter=200, warmup=150, seed=11); print("fit time", time.time() - t, flush=True)
dg = f.diagnostics()
print("nleap equal first 12:", (dg['n_leapfrog'][:, :12] == r.n_leapfrog[:, :12]).all(), "stepsize diff first 12", np.abs(dg['stepsize'][:, :12] - r.stepsize[:, :12]).max())
print("nleap equal frac", (dg['n_leapfrog'] == r.n_leapfrog).mean())
dr = f.draws()
print("draw mean diff (hypers)", dr[..., :3].mean((0, 1)) - r.draws[..., :3].mean((0, 1)))
tm = f.timing(); print("timing", tm, flush=True)
# ---- 3. PPC parity vs oracle on the GPU draws
ci, rngd = f.ppc(1.0, 0.05, 0.95, seed=5, return_counts_rng=True)
gq = O.generated_quantities(mo, dr.reshape(-1, dr.shape[-1]), 1.0, seed=5)
print("counts_rng identical frac", (gq == rngd).mean())
cio = O.summarise(gq, 0.05, 0.95)
print("ci max abs diff", np.abs(cio - ci).max())
f.close(); m.close()

# ---- 4. timing at 20k x 200
G, S, C = 20000, 200, 2
d = ind.synth(G, S, seed=20253, C=C)
K = d['K']
m = L.Model(d['counts'], d['X'], d['exposure'], K)
D = m.D
rng = np.random.default_rng(0)
u = rng.uniform(-0.5, 0.5, (4, D)); u[:, 3:3 + G] += 5
mo = O.model(d['counts'], d['X'], d['exposure'], K, n_threads=16)
t = time.time(); lpo, go = O.log_prob_grad(mo, u[0]); tc = time.time() - t
lp, g = m.log_prob_grad(u)
print("20kx200 lp rel err", abs(lp[0] - lpo) / abs(lpo), "grad err", np.max(np.abs(g[0] - go) / (1 + np.abs(go))), "cpu oracle s/grad (16 thr)", tc, flush=True)
for chains in [1, 4]:
    for Lg in [0, 8, 16, 32, 64]:
        m.set_launch(Lg, 0)
        t = time.time(); f = m.fit_nuts(chains=chains, iter=30, warmup=30, seed=1); el = time.time() - t
        tm = f.timing(); dg = f.diagnostics()
        print(f"chains {chains} L {m.get_launch()} wall {el:.3f}s pump {tm.seconds:.3f}s grad_evals {tm.grad_evals} us/grad {1e6*tm.seconds/max(tm.grad_evals,1):.2f} kA_ms {tm.gene_kernel_ms_mean:.4f} samples {tm.gene_kernel_samples} depth {dg['treedepth'].mean():.2f}", flush=True)
        f.close()
print("DONE")
