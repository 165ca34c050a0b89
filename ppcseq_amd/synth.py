"""Synthetic NB hierarchical count matrices (generator specified in SURVEY.md 8(d), configs 2-5).

The reference ships no generator; this one draws from the model of
inst/stan/negBinomial_MPI.stan itself (skew-normal intercepts :219, Laplace slopes :220,
sigma_raw regression :223, NB2-log likelihood :97-103) and injects outliers into some checked genes.
"""
from __future__ import annotations

import numpy as np


def synth(G, S, K=None, seed=20250, C=2, outliers=True):
    """Return dict(counts[G,S] int32, X[S,C], exposure[S], K, injected, truth)."""
    from scipy import stats
    rng = np.random.Generator(np.random.PCG64(seed))
    K = int(round(0.05 * G)) if K is None else int(K)
    group = np.zeros(S)
    group[(S + 1) // 2:] = 1.0
    cols = [np.ones(S), group]
    for _ in range(2, C):
        cols.append(rng.normal(size=S))
    X = np.stack(cols[:C], axis=1)
    exposure = rng.normal(0, 0.2, S)
    exposure -= exposure.mean()
    intercept = stats.skewnorm.rvs(-1.0, loc=6.5, scale=1.8, size=G, random_state=rng)
    sigma_raw = rng.normal(-0.3 * intercept, 0.4)
    phi = np.exp(-sigma_raw)
    alpha = np.zeros((C, G))
    alpha[0] = intercept
    if C >= 2:
        alpha[1, :K] = rng.laplace(0, 1, K)
    for c in range(2, C):
        alpha[c, :K] = rng.normal(0, 0.5, K)
    mu = np.exp((X @ alpha).T + exposure[None, :])
    lam = rng.gamma(phi[:, None], mu / phi[:, None])
    counts = rng.poisson(np.minimum(lam, 1e9)).astype(np.int64)
    injected = []
    if outliers and K > 0:
        n_out = max(1, K // 10)
        for g in rng.choice(K, n_out, replace=False):
            s = int(rng.integers(S))
            f = int(rng.integers(10, 51))
            up = (alpha[1, g] > 0) == (group[s] > 0.5) if C >= 2 else True
            counts[g, s] = counts[g, s] * f + f if up else counts[g, s] // f
            injected.append((int(g), s))
    counts = np.minimum(counts, 2**31 - 2).astype(np.int32)
    return dict(counts=counts, X=X, exposure=exposure, K=K, injected=injected,
                truth=dict(intercept=intercept, sigma_raw=sigma_raw, alpha=alpha))
