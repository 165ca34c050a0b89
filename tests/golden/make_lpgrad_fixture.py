"""Generates tests/golden/lpgrad_small.npz: small golden vectors for the log density and its gradient.

Inputs (counts, design, exposure, K, excluded cells, evaluation points) and expected outputs (lp, grad) of the model of
inst/stan/negBinomial_MPI.stan:180-258 on the unconstrained scale. The expected values are computed HERE with mpmath at
60 digits -- a direct transcription of the Stan program, gradient by central differences with h = 1e-25 -- so that they
are exact to double rounding also where fp64 formulas cancel (a count of 200 000: y - (y + phi) mu / (mu + phi));
they are cross-checked against the oracle (oracle/ppc_oracle.c), scipy.stats library densities (value) and torch fp64
autograd (value and gradient), oracle/independent.py, before they are written; the script refuses to write vectors on
which these disagree beyond their fp64 rounding.
The reference holds no numeric vectors for this path; these are the build's own, committed as data.
    python tests/golden/make_lpgrad_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import independent as ind  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

CASES = [  # G, S, C, K, seed, with exclusions
    (12, 7, 2, 3, 101, False), (20, 9, 2, 4, 102, True), (9, 5, 3, 2, 103, False), (8, 6, 1, 2, 104, True),
    (16, 21, 2, 3, 105, False)]

import mpmath as mp  # noqa: E402

mp.mp.dps = 60


def lp_mp(u, counts, X, expo, K, excl, lmm=mp.mpf("5.612671")):
    """Stan's target (inst/stan/negBinomial_MPI.stan:183-240): `~` statements without constants, lp_reduce with all terms."""
    G, S = counts.shape
    C = X.shape[1]
    o = ind.offsets(G, C, K)
    u = [mp.mpf(x) for x in u]
    lambda_mu = u[0] + lmm
    lambda_sigma = mp.e ** u[1]
    lambda_skew = u[2]
    icpt = u[o["intercept"]:o["intercept"] + G]
    a1 = u[o["alpha1"]:o["alpha1"] + K]
    a2 = u[o["alpha2"]:o["alpha2"] + (C - 2) * K] if C > 2 else []
    sraw = u[o["sigma_raw"]:o["sigma_raw"] + G]
    t = o["sigma_slope"]
    sigma_slope, sigma_intercept, sigma_sigma = -mp.e ** u[t], u[t + 1], mp.e ** u[t + 2]
    lp = u[1] + u[t] + u[t + 2]
    lp += -(lambda_mu - lmm) ** 2 / 8 - lambda_sigma ** 2 / 8 - lambda_skew ** 2 / 2
    lp += -sigma_intercept ** 2 / 8 - sigma_slope ** 2 / 8 - sigma_sigma ** 2 / 8
    xi = lambda_mu + lmm
    for g in range(G):
        z = (icpt[g] - xi) / lambda_sigma
        lp += -mp.log(lambda_sigma) - z * z / 2 + mp.log(mp.erfc(-lambda_skew * z / mp.sqrt(2)))
        r = sraw[g] - (sigma_slope * icpt[g] + sigma_intercept)
        lp += -mp.log(sigma_sigma) - r * r / (2 * sigma_sigma ** 2)
    if C >= 2:
        for g in range(K):
            lp += -abs(a1[g])
    for v in a2:
        lp += -v * v / (2 * mp.mpf("2.5") ** 2)
    ex = set(int(e) for e in excl)
    for g in range(G):
        phi = mp.e ** (-sraw[g])
        for s_ in range(S):
            if g * S + s_ in ex:
                continue
            eta = mp.mpf(float(expo[s_])) + mp.mpf(float(X[s_, 0])) * icpt[g]
            if g < K and C >= 2:
                eta += mp.mpf(float(X[s_, 1])) * a1[g]
                for c in range(2, C):
                    eta += mp.mpf(float(X[s_, c])) * a2[(c - 2) + (C - 2) * g]
            y = int(counts[g, s_])
            lp += (mp.loggamma(y + phi) - mp.loggamma(phi) - mp.loggamma(y + 1) + y * eta + phi * mp.log(phi)
                   - (y + phi) * mp.log(mp.e ** eta + phi))
    return lp


def lp_grad_mp(u, *a):
    lp = lp_mp(u, *a)
    h = mp.mpf("1e-25")
    g = np.zeros(len(u))
    for i in range(len(u)):
        up = [mp.mpf(x) for x in u]; um = list(up)
        up[i] += h; um[i] -= h
        g[i] = float((lp_mp(up, *a) - lp_mp(um, *a)) / (2 * h))
    return float(lp), g


O = Oracle()
out = {}
for n, (G, S, C, K, seed, with_excl) in enumerate(CASES):
    d = ind.synth(G, S, K=K, seed=seed, C=C)
    rng = np.random.default_rng(seed)
    counts = d["counts"].copy()
    counts[0, :] = rng.integers(0, 4, S)            # a low-count gene (exact-recurrence regime of the product)
    counts[1, 0] = 200_000                          # and one very large count (fp64 noise floor eps * y * eta ~ 3e-10)
    D = O.dim(G, C, K)
    u = rng.uniform(-1, 1, (3, D))
    u[:, 3:3 + G] += np.log(counts.mean(1) + 1.0)
    excl = np.array(sorted({1, S + 2, (G - 1) * S}), dtype=np.int32) if with_excl else np.zeros(0, np.int32)
    m = O.model(counts, d["X"], d["exposure"], K, excl=excl)
    lp = np.zeros(3)
    grad = np.zeros((3, D))
    for i in range(3):
        lp[i], grad[i] = lp_grad_mp(u[i], counts, d["X"], d["exposure"], K, excl)
        lp_o, g_o = O.log_prob_grad(m, u[i])
        lp_s = ind.log_prob_scipy(u[i], counts, d["X"], d["exposure"], K, excl=excl)
        lp_t, g_t = ind.log_prob_grad_torch(u[i], counts, d["X"], d["exposure"], K, excl=excl)
        assert abs(lp[i] - lp_o) <= 1e-11 * abs(lp_o), (n, i, lp[i], lp_o)   # the fp64 terms of the 2e5 count are ~3e6 each
        assert abs(lp[i] - lp_s) <= 1e-9 * abs(lp_s), (n, i, lp[i], lp_s)
        assert abs(lp[i] - lp_t) <= 1e-10 * abs(lp_t), (n, i, lp[i], lp_t)
        assert np.max(np.abs(grad[i] - g_o) / (1 + np.abs(g_o))) <= 1e-9, (n, i)     # the oracle's own fp64 cancellation
        assert np.max(np.abs(grad[i] - g_t) / (1 + np.abs(g_t))) <= 1e-8, (n, i)
        print("case", n, "point", i, "lp", lp[i], "oracle grad err", np.max(np.abs(grad[i] - g_o) / (1 + np.abs(g_o))), flush=True)
    for k, v in dict(counts=counts, X=d["X"], exposure=d["exposure"], K=np.array(K), excl=excl, u=u, lp=lp, grad=grad).items():
        out[f"c{n}_{k}"] = v
out["n_cases"] = np.array(len(CASES))
np.savez_compressed(os.path.join(HERE, "lpgrad_small.npz"), **out)
print("wrote", os.path.join(HERE, "lpgrad_small.npz"), "cases:", len(CASES))
