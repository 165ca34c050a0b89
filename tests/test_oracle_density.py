"""Pins the C oracle (oracle/ppc_oracle.c) against independent library evaluations of the same Stan
model (oracle/independent.py): scipy.stats densities for the value, torch fp64 autograd and central
finite differences for the gradient, mpmath for the special functions at extreme arguments.
The reference holds no numeric golden vectors for this path, so this IS the pin."""
import math

import numpy as np
import pytest

from oracle import independent as ind

CASES = [(7, 5, 2, 3, 1), (40, 21, 2, 5, 2), (30, 11, 3, 4, 3), (12, 6, 1, 2, 4), (25, 9, 5, 6, 5), (9, 1, 2, 2, 6), (1, 4, 2, 1, 7)]


def _point(G, S, C, K, seed, oracle):
    d = ind.synth(G, S, K=K, seed=seed, C=C)
    rng = np.random.default_rng(seed)
    D = oracle.dim(G, C, K)
    u = rng.uniform(-1, 1, D)
    u[3:3 + G] += 5
    excl = np.array(sorted({1 % (G * S), (2 * S + 3) % (G * S), (G - 1) * S}), dtype=np.int32) if seed % 2 == 0 else None
    return d, u, excl


@pytest.mark.parametrize("G,S,C,K,seed", CASES)
def test_value_matches_scipy(oracle, G, S, C, K, seed):
    d, u, excl = _point(G, S, C, K, seed, oracle)
    m = oracle.model(d["counts"], d["X"], d["exposure"], K, excl=excl)
    lp, _ = oracle.log_prob_grad(m, u)
    ref = ind.log_prob_scipy(u, d["counts"], d["X"], d["exposure"], K, excl=excl)
    assert abs(lp - ref) <= 1e-9 * max(1.0, abs(ref))


@pytest.mark.parametrize("G,S,C,K,seed", CASES)
def test_gradient_matches_autograd(oracle, G, S, C, K, seed):
    d, u, excl = _point(G, S, C, K, seed, oracle)
    m = oracle.model(d["counts"], d["X"], d["exposure"], K, excl=excl)
    lp, g = oracle.log_prob_grad(m, u)
    lp_t, g_t = ind.log_prob_grad_torch(u, d["counts"], d["X"], d["exposure"], K, excl=excl)
    assert abs(lp - lp_t) <= 1e-9 * max(1.0, abs(lp_t))
    assert np.max(np.abs(g - g_t) / (1 + np.abs(g_t))) < 1e-10


def test_gradient_matches_finite_differences(oracle):
    d, u, excl = _point(15, 7, 3, 4, 11, oracle)
    m = oracle.model(d["counts"], d["X"], d["exposure"], 4, excl=excl)
    _, g = oracle.log_prob_grad(m, u)
    rng = np.random.default_rng(0)
    for _ in range(5):
        v = rng.normal(size=u.size)
        v /= np.linalg.norm(v)
        h = 1e-5
        fd = (oracle.log_prob_grad(m, u + h * v, False)[0] - oracle.log_prob_grad(m, u - h * v, False)[0]) / (2 * h)
        assert abs(fd - g @ v) < 1e-5 * max(1.0, abs(fd))


def test_digamma_against_mpmath(oracle):
    import mpmath as mp
    mp.mp.dps = 40
    for x in [1e-3, 0.1, 0.5, 1.0, 2.5, 7.99, 8.0, 10.0, 123.456, 2.6e6, 1e9]:
        ref = float(mp.digamma(mp.mpf(x)))
        assert abs(oracle.lib.ppco_digamma(x) - ref) <= 2e-14 * max(1.0, abs(ref))


def test_extreme_counts_and_dispersion(oracle):
    """y = 0, y = 2.58e6 (the bundled maximum), phi from 1e-3 to 1e5: value agrees with mpmath."""
    import mpmath as mp
    mp.mp.dps = 50
    counts = np.array([[0, 2580228, 7, 100000]], dtype=np.int32)
    X = np.ones((4, 1))
    expo = np.array([0.1, -0.2, 0.0, 0.3])
    for sigma_raw in [math.log(1e3), 0.0, -math.log(1e5)]:
        for intercept in [-5.0, 3.0, 14.0]:
            u = np.array([0.3, -0.2, 0.1, intercept, sigma_raw, -0.5, 0.2, -0.7])
            m1 = oracle.model(counts, X, expo, 0)
            m0 = oracle.model(counts, X, expo, 0, excl=np.arange(4, dtype=np.int32))
            lik = oracle.log_prob_grad(m1, u)[0] - oracle.log_prob_grad(m0, u)[0]
            phi = mp.e ** (-mp.mpf(sigma_raw))
            ref = mp.mpf(0)
            for s in range(4):
                y = int(counts[0, s])
                eta = mp.mpf(float(expo[s])) + mp.mpf(intercept)
                ref += (mp.loggamma(y + phi) - mp.loggamma(phi) - mp.loggamma(y + 1) + y * eta + phi * mp.log(phi)
                        - (y + phi) * mp.log(mp.e ** eta + phi))
            assert abs(lik - float(ref)) <= 1e-9 * max(1.0, abs(float(ref)))
