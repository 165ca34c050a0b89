"""The per-gene dispersion tables (ppcseq_amd/csrc/ppcx_disp.h) against mpmath: the count-and-dispersion part of
neg_binomial_2_log_lpmf (inst/stan/negBinomial_MPI.stan:97-103) summed over a gene's cells,
    Fh(sigma) = sum_s lgamma(y_s + phi) - lgamma(phi) + y_s sigma + y_s,   Dh(sigma) = sum_s psi(y_s + phi) - psi(phi),   phi = exp(-sigma),
read from the table inside its range and evaluated directly outside. Runs the product's headers on the host (tests/emul)."""
import ctypes as C

import mpmath as mp
import numpy as np
import pytest

from tests.emul_util import P


def _rows():
    rng = np.random.default_rng(7)
    S = 60
    rows = {
        "low": rng.negative_binomial(2.0, 2.0 / (2.0 + 5.0), S),
        "mid": rng.negative_binomial(7.0, 7.0 / (7.0 + 600.0), S),
        "high": rng.negative_binomial(3.0, 3.0 / (3.0 + 2e5), S),
        "zeros": np.zeros(S, dtype=np.int64),
        "mixed": np.concatenate([np.arange(0, 20), rng.integers(0, 3_000_000, 40)]),
    }
    ex = rows["mid"].copy(); ex[[3, 17, 40]] = -1                 # excluded cells (to_exclude, R/utilities.R:321-359)
    rows["excluded"] = ex
    return {k: np.ascontiguousarray(v, np.int32) for k, v in rows.items()}


def _exact(row, sigma):
    mp.mp.dps = 40
    phi = mp.e ** (-mp.mpf(sigma))
    F = mp.mpf(0); D = mp.mpf(0)
    for y in row:
        if y < 0:
            continue
        y = int(y)
        F += mp.loggamma(y + phi) - mp.loggamma(phi) + y * mp.mpf(sigma) + y
        D += mp.digamma(y + phi) - mp.digamma(phi)
    return F, D


@pytest.mark.parametrize("name", ["low", "mid", "high", "zeros", "mixed", "excluded"])
def test_table_and_direct_evaluation_match_mpmath(emul, name):
    row = _rows()[name]
    rng = np.random.default_rng(11)
    # inside the range (panel edges and interiors), at its ends, and outside (direct evaluation)
    sig = np.concatenate([rng.uniform(-8, 8, 24), [-8.0, 7.999999, -7.5, 0.0, 0.5 - 1e-12, 0.5, 3.25],
                          [-11.0, -8.000001, 8.0, 9.5, 14.0]])
    n = sig.size
    F = np.zeros(n); D = np.zeros(n); Fd = np.zeros(n); Dd = np.zeros(n); inr = np.zeros(n, np.int32)
    emul.emul_disp_table(P(row, C.c_int32), int(row.size), n, P(sig, C.c_double), P(F, C.c_double), P(D, C.c_double),
                         P(Fd, C.c_double), P(Dd, C.c_double), P(inr, C.c_int))
    assert inr[:31].all() and not inr[31:].any()
    for i in range(n):
        eF, eD = _exact(row, float(sig[i]))
        # scale: the sum of the cells' magnitudes (every cell's term has one sign), a few units of rounding of it
        mp.mp.dps = 40
        for got, gd, ex in ((F[i], Fd[i], eF), (D[i], Dd[i], eD)):
            tol = 8e-16 * float(abs(ex)) + 1e-13
            assert abs(got - float(ex)) <= tol, (name, sig[i], got, float(ex))
            assert abs(gd - float(ex)) <= tol, (name, "direct", sig[i], gd, float(ex))
