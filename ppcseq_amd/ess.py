"""Convergence / efficiency diagnostics computed the same way for the GPU and the CPU runs.

The reference discards n_eff and Rhat (R/utilities.R:699), so the headline metric "effective samples
per second" needs its own estimator: rank-normalised split-chain bulk-ESS and Rhat of Vehtari, Gelman,
Simpson, Carpenter, Buerkner (2021), "Rank-normalization, folding, and localization".
"""
from __future__ import annotations

import numpy as np


def _split(x):
    n = x.shape[1] // 2
    return np.concatenate([x[:, :n], x[:, x.shape[1] - n:]], axis=0)


def _rank_normalise(x):
    from scipy import stats
    r = stats.rankdata(x.reshape(-1), method="average").reshape(x.shape)
    return stats.norm.ppf((r - 0.375) / (x.size + 0.25))


def _autocov(x):
    n = x.shape[1]
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=1, keepdims=True)
    f = np.fft.rfft(xc, m, axis=1)
    ac = np.fft.irfft(f * np.conj(f), m, axis=1)[:, :n]
    return ac / n


def _ess_raw(x):
    """x: [chains, draws] -> ESS by Geyer's initial monotone sequence over the chain-averaged autocovariance."""
    m, n = x.shape
    if n < 4:
        return float("nan")
    acov = _autocov(x)
    chain_var = acov[:, 0] * n / (n - 1.0)
    mean_var = chain_var.mean()
    var_plus = mean_var * (n - 1.0) / n
    if m > 1:
        var_plus += x.mean(axis=1).var(ddof=1)
    if not var_plus > 0:
        return float("nan")
    rho = 1.0 - (mean_var - acov.mean(axis=0)) / var_plus
    rho[0] = 1.0
    prev = np.inf
    pairs = []
    # Geyer's initial monotone sequence over the pairs (rho0+rho1), (rho2+rho3), ...
    k = 0
    while 2 * k + 1 < n:
        p = rho[2 * k] + rho[2 * k + 1]
        if p < 0:
            break
        p = min(p, prev)
        pairs.append(p)
        prev = p
        k += 1
    tau = -1.0 + 2.0 * sum(pairs)
    tau = max(tau, 1.0 / np.log10(m * n))
    return m * n / tau


def ess_bulk(x):
    """Rank-normalised split-chain bulk ESS of draws x[chains, draws]."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    return _ess_raw(_rank_normalise(_split(x)))


def rhat(x):
    """Rank-normalised split-chain Rhat (bulk)."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    z = _rank_normalise(_split(x))
    n = z.shape[1]
    w = z.var(axis=1, ddof=1).mean()
    b = n * z.mean(axis=1).var(ddof=1)
    return float(np.sqrt(((n - 1.0) / n * w + b / n) / w))


def summary_ess(draws, lp, hyper_cols):
    """Headline ESS of a fit: min bulk-ESS over the six hyper-parameters and lp__.

    draws: [chains, n_keep, n_cols] (columns = hyper_cols order); lp: [chains, n_keep].
    """
    vals = [ess_bulk(draws[:, :, j]) for j in range(len(hyper_cols))] + [ess_bulk(lp)]
    return dict(min=float(np.nanmin(vals)), per_param=[float(v) for v in vals])
