"""Host-logic tests without a GPU: the product's `__host__ __device__` arithmetic and NUTS state machine
(ppcseq_amd/csrc/ppcx_{math,model,nuts,gene}.h), compiled for the CPU by tests/emul/ppcx_emul.cpp with
loops in place of wavefront lanes, against the oracle. The GPU kernels themselves are checked by the
`-m gpu` tests through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from oracle import independent as ind
from tests.emul_util import emul_fit, emul_lp

CASES = [(7, 5, 2, 3, 1), (40, 21, 2, 5, 2), (30, 11, 3, 4, 3), (12, 6, 1, 2, 4), (25, 9, 5, 6, 5), (9, 1, 2, 2, 6), (6, 3, 2, 0, 8)]


@pytest.mark.parametrize("G,S,C,K,seed", CASES)
def test_density_and_gradient(oracle, emul, G, S, C, K, seed):
    d = ind.synth(G, S, K=K, seed=seed, C=C)
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1, 1, oracle.dim(G, C, K))
    u[3:3 + G] += 5
    excl = np.array(sorted({1 % (G * S), (2 * S + 3) % (G * S), (G - 1) * S}), dtype=np.int32) if seed % 2 == 0 else None
    m = oracle.model(d["counts"], d["X"], d["exposure"], K, excl=excl)
    lp, g = oracle.log_prob_grad(m, u)
    lp2, g2 = emul_lp(emul, d["counts"], d["X"], d["exposure"], K, u, excl)
    assert abs(lp2 - lp) <= 1e-11 * max(1.0, abs(lp))
    assert np.max(np.abs(g - g2) / (1 + np.abs(g))) < 1e-10


@pytest.mark.parametrize("levels,G,S,K,seed", [((3,), 30, 14, 6, 1), ((2, 2), 24, 11, 24, 2), ((4,), 18, 9, 5, 3), ((3, 2), 20, 13, 7, 4),
                                               ((12,), 26, 40, 7, 5)])
def test_factor_designs_take_the_factorised_cells(oracle, emul, levels, G, S, K, seed):
    """Designs whose slope columns are all 0 / 1 indicators -- a multi-level factor, `~ a + b` of factors (model.matrix,
    R/utilities.R:887-900) -- with C = 3, 3 and 4, 4: e^t = E_s A_g prod exp(slope_c), no exp per cell (ppcx_gene.h
    indicator_cells). Low counts and excluded cells included."""
    d = ind.synth_factor(G, S, K, levels, seed)
    counts = d["counts"].copy()
    counts[1, :] //= 50; counts[2, ::2] = 0                              # list cells (y <= 7) in genes with slopes
    C = d["X"].shape[1]
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1, 1, oracle.dim(G, C, K))
    u[3:3 + G] += 5
    excl = np.array(sorted({1, S + 2, 2 * S, (G - 1) * S + 1}), dtype=np.int32) if seed % 2 == 0 else None
    m = oracle.model(counts, d["X"], d["exposure"], K, excl=excl)
    lp, g = oracle.log_prob_grad(m, u)
    lp2, g2 = emul_lp(emul, counts, d["X"], d["exposure"], K, u, excl)
    assert abs(lp2 - lp) <= 1e-11 * max(1.0, abs(lp))
    assert np.max(np.abs(g - g2) / (1 + np.abs(g))) < 1e-10


def _out_of_range_case(seed):
    """sigma_raw beyond both ends of the genes' dispersion tables (ppcx_disp.h: [-8, 8)) -- phi = exp(-sigma_raw) above 2981 and
    below 3.4e-4, where a warm-up trajectory can stray -- beside positions inside, at a panel edge and at the ends themselves;
    counts on both sides of the y = 8 switch of the direct evaluation, a two-group design, excluded cells."""
    rng = np.random.default_rng(seed)
    G, S, K = 24, 13, 5
    counts = rng.poisson(rng.choice([0.5, 5.0, 40.0, 3000.0], size=(G, 1)), size=(G, S)).astype(np.int32)
    X = np.stack([np.ones(S), (np.arange(S) % 2).astype(float)], axis=1)
    expo = rng.normal(0, 0.2, S)
    u = rng.uniform(-1, 1, 2 * G + K + 6)
    u[3:3 + G] = np.log(counts.mean(1) + 0.5) + rng.normal(0, 0.3, G)
    sr = rng.uniform(-7.9, 7.9, G)
    sr[:10] = [-12.5, -8.000001, -8.0, 8.0, 8.3, 11.0, 7.999999, 0.5, -0.5, 16.0]
    u[3 + G + K:3 + G + K + G] = sr
    excl = np.array([2, 3 * S + 1, 5 * S + 4, 9 * S], dtype=np.int32)
    return counts, X, expo, K, u, excl


@pytest.mark.parametrize("seed", [1, 2])
def test_positions_outside_the_tabulated_dispersion_range(oracle, emul, seed):
    counts, X, expo, K, u, excl = _out_of_range_case(seed)
    for ex in (None, excl):
        m = oracle.model(counts, X, expo, K, excl=ex)
        lp, g = oracle.log_prob_grad(m, u)
        lp2, g2 = emul_lp(emul, counts, X, expo, K, u, ex)
        assert abs(lp2 - lp) <= 1e-11 * max(1.0, abs(lp))
        assert np.max(np.abs(g - g2) / (1 + np.abs(g))) < 1e-10


def _low_count_case(seed):
    """Counts 0..40 around the regime boundaries of the cell loop (y + phi < 8: exact recurrences; < 32: 7-term tails;
    else 4-term) with phi from 0.01 to 100, some rows all zero, some cells excluded."""
    rng = np.random.default_rng(seed)
    G, S, K = 48, 23, 6
    counts = rng.poisson(rng.choice([0.3, 2.0, 6.0, 9.0, 30.0], size=(G, 1)), size=(G, S)).astype(np.int32)
    counts[5] = 0
    counts[7, :] = 7
    counts[8, :] = 8
    X = np.stack([np.ones(S), (np.arange(S) % 2).astype(float)], axis=1)
    expo = rng.normal(0, 0.2, S)
    D = 2 * G + K + 6
    u = rng.uniform(-1, 1, D)
    u[3:3 + G] = np.log(counts.mean(1) + 0.5) + rng.normal(0, 0.3, G)
    u[3 + G + K:3 + G + K + G] = rng.uniform(-4.6, 4.6, G)          # sigma_raw: phi = exp(-sigma_raw) in (0.01, 100)
    excl = np.array([3, 5 * S + 1, 7 * S, 8 * S + 2], dtype=np.int32)
    return counts, X, expo, K, u, excl


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_low_counts_across_the_lgamma_regimes(oracle, emul, seed):
    counts, X, expo, K, u, excl = _low_count_case(seed)
    m = oracle.model(counts, X, expo, K, excl=excl)
    lp, g = oracle.log_prob_grad(m, u)
    lp2, g2 = emul_lp(emul, counts, X, expo, K, u, excl)
    assert abs(lp2 - lp) <= 1e-11 * max(1.0, abs(lp))
    assert np.max(np.abs(g - g2) / (1 + np.abs(g))) < 1e-10


def test_design_without_unit_intercept_column(oracle, emul):
    """X[,1] != 1 disables the E_s * A_g factorisation: the generic per-cell exp path must agree too."""
    d = ind.synth(10, 6, K=3, seed=2, C=2)
    X = d["X"].copy()
    X[:, 0] = np.linspace(0.5, 1.5, 6)
    u = np.random.default_rng(1).uniform(-1, 1, oracle.dim(10, 2, 3))
    u[3:13] += 4
    m = oracle.model(d["counts"], X, d["exposure"], 3)
    lp, g = oracle.log_prob_grad(m, u)
    lp2, g2 = emul_lp(emul, d["counts"], X, d["exposure"], 3, u)
    assert abs(lp2 - lp) <= 1e-11 * abs(lp) and np.max(np.abs(g - g2) / (1 + np.abs(g))) < 1e-10


def test_continuous_covariate_keeps_the_per_cell_exp(oracle, emul):
    """C = 2 with a unit intercept column but a continuous second column: not a two-group design, so the checked genes
    take the generic cell path while the other genes of the same wavefronts take the factorised one."""
    d = ind.synth(21, 9, K=5, seed=6, C=2)
    X = d["X"].copy()
    X[:, 1] = np.linspace(-1.0, 1.5, 9)
    u = np.random.default_rng(2).uniform(-1, 1, oracle.dim(21, 2, 5))
    u[3:24] += 4
    m = oracle.model(d["counts"], X, d["exposure"], 5)
    lp, g = oracle.log_prob_grad(m, u)
    lp2, g2 = emul_lp(emul, d["counts"], X, d["exposure"], 5, u)
    assert abs(lp2 - lp) <= 1e-11 * abs(lp) and np.max(np.abs(g - g2) / (1 + np.abs(g))) < 1e-10


@pytest.mark.parametrize("G,S,C,K,seed", [(30, 8, 2, 4, 3), (20, 6, 1, 3, 4), (24, 7, 3, 4, 5)])
def test_nuts_state_machine_follows_oracle(oracle, emul, G, S, C, K, seed):
    """Same Philox streams, same Stan-default algorithm: the iterative device tree and the oracle's
    recursive build_tree must make identical decisions until floating-point chaos separates the chains.
    The first iterations (before roundoff has been amplified by the long warmup trajectories) agree to
    rounding, including trees up to depth >= 5."""
    d = ind.synth(G, S, K=K, seed=seed, C=C)
    m = oracle.model(d["counts"], d["X"], d["exposure"], K)
    r = oracle.nuts_model(m, oracle.cfg(chains=2, iter=40, warmup=40, seed=11))
    e = emul_fit(emul, d["counts"], d["X"], d["exposure"], K, 2, 40, 40, 11)
    n = 12
    assert np.array_equal(r.n_leapfrog[:, :n], e["n_leapfrog"][:, :n])
    assert np.array_equal(r.treedepth[:, :n], e["treedepth"][:, :n])
    assert np.array_equal(r.divergent[:, :n], e["divergent"][:, :n])
    assert np.max(np.abs(r.stepsize[:, :n] - e["stepsize"][:, :n])) < 1e-9
    assert np.max(np.abs(r.accept[:, :n] - e["accept"][:, :n])) < 1e-7
    assert r.treedepth[:, :n].max() >= 4


def test_nuts_sampling_draws_follow_oracle(oracle, emul):
    """With warmup = 0 (no adaptation, unit metric) and a capped tree depth the kept draws themselves
    can be compared for the first iterations."""
    d = ind.synth(16, 5, K=3, seed=9, C=2)
    m = oracle.model(d["counts"], d["X"], d["exposure"], 3)
    cfg = oracle.cfg(chains=1, iter=8, warmup=0, seed=5, max_treedepth=6)
    r = oracle.nuts_model(m, cfg)
    e = emul_fit(emul, d["counts"], d["X"], d["exposure"], 3, 1, 8, 0, 5, max_treedepth=6)
    assert np.array_equal(r.n_leapfrog, e["n_leapfrog"])
    assert np.max(np.abs(r.draws - e["draws"])) < 1e-8
    assert np.max(np.abs(r.lp - e["lp"])) < 1e-7


def test_full_run_distribution_matches_oracle(oracle, emul):
    """Over a whole run the two samplers target the same posterior: pooled means of the hyper-parameters
    agree within Monte-Carlo error."""
    d = ind.synth(40, 10, K=4, seed=21, C=2)
    m = oracle.model(d["counts"], d["X"], d["exposure"], 4)
    r = oracle.nuts_model(m, oracle.cfg(chains=4, iter=400, warmup=150, seed=3))
    e = emul_fit(emul, d["counts"], d["X"], d["exposure"], 4, 4, 400, 150, 3)
    D = r.draws.shape[-1]
    cols = [0, 1, 2, D - 3, D - 2, D - 1]
    a = r.draws[..., cols].reshape(-1, 6)
    b = e["draws"][..., cols].reshape(-1, 6)
    se = np.sqrt(a.var(0) / 100 + b.var(0) / 100)             # ESS >= 100 each, conservatively
    assert np.all(np.abs(a.mean(0) - b.mean(0)) < 5 * se)
    assert e["divergent"][:, 150:].mean() <= 0.02       # small hierarchical model: rare divergences are expected


def test_nb_rng_spec_identical(oracle, emul):
    """The product's gamma-Poisson generator and the oracle's are two implementations of one Philox
    stream specification: identical integers."""
    emul.emul_nb2_log_rng.restype = C.c_int
    emul.emul_nb2_log_rng.argtypes = [C.c_double, C.c_double, C.c_ulonglong, C.c_uint, C.c_uint]
    rng = np.random.default_rng(0)
    for _ in range(300):
        eta, phi = rng.uniform(-3, 12), np.exp(rng.uniform(-4, 6))
        cell, draw, seed = int(rng.integers(1 << 20)), int(rng.integers(1 << 16)), int(rng.integers(1 << 40))
        assert emul.emul_nb2_log_rng(eta, phi, seed, cell, draw) == oracle.nb2_log_rng(eta, phi, seed, cell, draw)


def test_golden_vectors(oracle, emul):
    """Committed golden vectors (tests/golden/lpgrad_small.npz: mpmath at 60 digits, cross-checked with the oracle, scipy
    and torch autograd by make_lpgrad_fixture.py): the oracle agrees to its own fp64 rounding (it cancels y - (y + phi)
    mu / (mu + phi) at the 200 000 count) and the product's arithmetic (CPU emulation) to the parity tolerances."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lpgrad_small.npz"))
    for n in range(int(z["n_cases"])):
        g = {k: z[f"c{n}_{k}"] for k in ("counts", "X", "exposure", "K", "excl", "u", "lp", "grad")}
        K = int(g["K"])
        m = oracle.model(g["counts"], g["X"], g["exposure"], K, excl=g["excl"])
        for i in range(g["u"].shape[0]):
            lp, gr = oracle.log_prob_grad(m, g["u"][i])
            assert abs(lp - g["lp"][i]) <= 1e-11 * abs(g["lp"][i]) and np.max(np.abs(gr - g["grad"][i]) / (1 + np.abs(gr))) < 1e-9
            lp2, g2 = emul_lp(emul, g["counts"], g["X"], g["exposure"], K, g["u"][i], g["excl"] if g["excl"].size else None)
            assert abs(lp2 - g["lp"][i]) <= 1e-11 * max(1.0, abs(g["lp"][i]))
            assert np.max(np.abs(g2 - g["grad"][i]) / (1 + np.abs(g["grad"][i]))) < 1e-10


def test_approximated_analysis_oracle_properties(oracle):
    """The oracle's restatement of fit_to_counts_rng_approximated (R/utilities.R:733-784): resampling indices are uniform
    over the posterior draws, a draw equals neg_binomial_2_log_rng on the resampled parameters, and with one posterior
    draw the approximated and the full analysis see the same parameters."""
    d = ind.synth(10, 6, K=3, seed=4)
    m = oracle.model(d["counts"], d["X"], d["exposure"], 3)
    D = oracle.dim(10, 2, 3)
    rng = np.random.default_rng(0)
    dr = rng.normal(0, 0.2, (50, D)); dr[:, 3:13] += 5
    a = oracle.generated_quantities_approx(m, dr, 4000, 0.7352941, seed=9)
    assert a.shape == (4000, 3, 6) and a.min() >= 0
    full = oracle.generated_quantities(m, dr, 0.7352941, seed=9)
    # same predictive distribution: means of the two estimates agree within Monte-Carlo error of the heavy-tailed draws
    assert np.all(np.abs(a.mean(0) / full.mean(0) - 1) < 0.5)
    one = oracle.generated_quantities_approx(m, dr[:1], 7, 1.0, seed=3)
    ref = np.stack([[[oracle.nb2_log_rng(d["exposure"][s] + dr[0, 3 + g] + d["X"][s, 1] * dr[0, 13 + g], np.exp(-dr[0, 16 + g]), 3, g * 6 + s, j)
                      for s in range(6)] for g in range(3)] for j in range(7)])
    assert np.array_equal(one, ref)


@pytest.mark.parametrize("G,S,C,K,seed", [(30, 8, 2, 4, 3), (20, 6, 1, 3, 4), (24, 7, 3, 4, 5)])
def test_pipelined_rounds_reproduce_the_classic_rounds(emul, G, S, C, K, seed):
    """The two-launch round (log-likelihood beside the state machine, positions anticipated by the gene kernel, a carried
    round when the anticipation was wrong) is a re-ordering of the same arithmetic: the same chains come out, whichever
    of the two concurrent parts of the merged launch runs first, with and without anticipation. The only difference is
    the order in which the kinetic energy of fresh momenta is summed (per gene instead of per coordinate index), so the
    early iterations agree to rounding."""
    from tests.emul_util import emul_fit_pipelined
    d = ind.synth(G, S, K=K, seed=seed, C=C)
    if C == 3:
        d["X"][:, 2] = np.linspace(-1, 1, S)              # a continuous covariate: genes with the per-cell-eta path
    e = emul_fit(emul, d["counts"], d["X"], d["exposure"], K, 2, 60, 40, 11)
    n = 14
    total_leaps = e["n_leapfrog"].sum(1)
    # genes with the per-cell-eta path (C == 3 here) read the coefficients kept among the coordinates' constants (round 5:
    # Dims::raw_consts), so such a model anticipates and pipelines like every other
    generic = C == 3
    for spec in (True, False):
        for first_s in (False, True):
            p = emul_fit_pipelined(emul, d["counts"], d["X"], d["exposure"], K, 2, 60, 40, 11, spec=spec, ls_first_s=first_s)
            assert np.array_equal(p["n_leapfrog"][:, :n], e["n_leapfrog"][:, :n])
            assert np.array_equal(p["treedepth"][:, :n], e["treedepth"][:, :n])
            assert np.max(np.abs(p["stepsize"][:, :n] - e["stepsize"][:, :n])) < 1e-9
            assert np.max(np.abs(p["accept"][:, :n] - e["accept"][:, :n])) < 1e-7
            if first_s:
                ps = p
            leaps = p["n_leapfrog"].sum(1)
            if spec:
                # a carried round per mis-anticipation: new transitions, direction changes, the step-size search
                assert np.all(p["carried"] < 0.35 * leaps + 200), (p["carried"], leaps)
            else:
                assert np.all(p["carried"] >= leaps)          # nothing anticipated: every evaluation takes two rounds
    # the two orders of the merged launch give the same bits
    p2 = emul_fit_pipelined(emul, d["counts"], d["X"], d["exposure"], K, 2, 60, 40, 11, spec=True, ls_first_s=False)
    assert np.array_equal(p2["draws"], ps["draws"]) and np.array_equal(p2["n_leapfrog"], ps["n_leapfrog"])
    assert total_leaps.min() > 100
