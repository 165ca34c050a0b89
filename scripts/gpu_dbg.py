import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppcseq_amd import _lib as L
from oracle.oracle import Oracle
from oracle import independent as ind
import mpmath as mp
O = Oracle()
G, S, C, K, seed = 257, 200, 2, 13, 7
d = ind.synth(G, S, K=K, seed=seed, C=C)
rng = np.random.default_rng(seed)
D = O.dim(G, C, K)
u = rng.uniform(-1, 1, (3, D)); u[:, 3:3 + G] += 5
mo = O.model(d['counts'], d['X'], d['exposure'], K)
m = L.Model(d['counts'], d['X'], d['exposure'], K)
for Lg in [0, 1, 16]:
    m.set_launch(Lg, 0)
    lp, g = m.log_prob_grad(u)
    for i in range(3):
        lpo, go = O.log_prob_grad(mo, u[i])
        print(Lg, i, lpo, lp[i] - lpo, np.max(np.abs(g[i] - go)), np.argmax(np.abs(g[i]-go)))
# high-precision likelihood for point 0 with mpmath (likelihood part only differs between implementations)
mp.mp.dps = 40
ui = u[0]
off_sr = 3 + G + K
tot = mp.mpf(0)
for gidx in range(G):
    phi = mp.exp(-mp.mpf(ui[off_sr + gidx]))
    for s in range(S):
        eta = mp.mpf(d['exposure'][s]) + mp.mpf(ui[3 + gidx]) + (mp.mpf(d['X'][s, 1]) * mp.mpf(ui[3 + G + gidx]) if gidx < K else 0)
        y = int(d['counts'][gidx, s])
        tot += mp.loggamma(y + phi) - mp.loggamma(phi) - mp.loggamma(y + 1) + y * eta + phi * mp.log(phi) - (y + phi) * mp.log(mp.exp(eta) + phi)
# priors via oracle with zero-likelihood trick: evaluate oracle on counts but subtract its own likelihood is messy; instead compare likelihood-only by differencing two models
print("mp likelihood", tot)
import copy
# oracle likelihood = oracle lp - priors; priors = lp of model with all cells excluded
excl_all = np.arange(G * S, dtype=np.int32)
mo0 = O.model(d['counts'], d['X'], d['exposure'], K, excl=excl_all)
pri, _ = O.log_prob_grad(mo0, ui)
lpo, _ = O.log_prob_grad(mo, ui)
m.set_launch(0, 0)
lpg, _ = m.log_prob_grad(ui)
print("oracle lik err", float(mp.mpf(lpo) - mp.mpf(pri) - tot), "gpu lik err", float(mp.mpf(lpg) - mp.mpf(pri) - tot))
