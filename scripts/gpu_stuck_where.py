"""Development aid: where does a chain that left warm-up at tree depth 10 sit? Chains CHAIN (the stuck one) and
CHAIN + 1 of the cfg3 fit at SEED, the last kept draws side by side: the coordinates in which the stuck chain is furthest from the other
chain, in units of the other chain's spread, and the counts of those genes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from ppcseq_amd.synth import synth
d = synth(20000, 200, seed=20253)
G, K = 20000, d["K"]
m = L.Model(d["counts"], d["X"], d["exposure"], K)
m.set_launch(int(os.environ.get("LANES", 8)), 0)
it = int(os.environ.get("ITER", 190))
f = m.fit_nuts(chains=2, iter=it, warmup=150, seed=int(os.environ.get("SEED", 2)), chain_id_offset=int(os.environ.get("CHAIN", 0)))
dr = f.draws()
dg = f.diagnostics()
print("step sizes", dg["stepsize"][:, -1], "depths", dg["treedepth"][:, -1])
bad, good = dr[0], dr[1]
mu, sd = good.mean(0), good.std(0) + 1e-12
z = (bad[-1] - mu) / sd
names = (["lambda_mu", "lambda_sigma", "lambda_skew"] + [f"intercept[{g}]" for g in range(G)] + [f"alpha[{g}]" for g in range(K)]
         + [f"sigma_raw[{g}]" for g in range(G)] + ["sigma_slope", "sigma_intercept", "sigma_sigma"])
order = np.argsort(-np.abs(z))[:12]
for i in order:
    print(f"{names[i]:20s} stuck {bad[-1, i]:10.4f}  (sd within the stuck chain {bad[:, i].std():.2e})  other chain {mu[i]:10.4f} +- {sd[i]:.2e}   z {z[i]:9.1f}")
for i in order[:4]:
    nm = names[i]
    if "[" in nm:
        g = int(nm[nm.index("[") + 1:-1])
        y = d["counts"][g]
        print(nm, "counts: min", y.min(), "median", int(np.median(y)), "max", y.max(), "mean", y.mean().round(1), "var/mean", (y.var() / max(y.mean(), 1e-9)).round(2),
              "| stuck intercept", bad[-1, 3 + g], "sigma_raw", bad[-1, 3 + G + K + g], "| other intercept", mu[3 + g], "sigma_raw", mu[3 + G + K + g])
        print("   first 40 counts", y[:40].tolist())
# the metric each chain ended warm-up with (ppcx_fit_get_inv_metric): where is the stuck chain's far from the other's?
im = f.inv_metric()
r = im[0] / im[1]
print("inverse metric, stuck / other: percentiles 0 1 50 99 100", np.percentile(r, [0, 1, 50, 99, 100]))
print("inverse metric / variance of the other chain's kept draws: stuck", np.percentile(im[0] / good.var(0), [0, 1, 50, 99, 100]), "other", np.percentile(im[1] / good.var(0), [0, 1, 50, 99, 100]))
for i in np.argsort(-r)[:10]:
    print(f"{names[i]:20s} inv metric stuck {im[0, i]:.3e} other {im[1, i]:.3e}  variance of the other chain's draws {good.var(0)[i]:.3e}")
for i in np.argsort(r)[:5]:
    print(f"{names[i]:20s} inv metric stuck {im[0, i]:.3e} other {im[1, i]:.3e}  variance of the other chain's draws {good.var(0)[i]:.3e}")
