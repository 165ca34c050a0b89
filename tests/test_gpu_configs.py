"""GPU parity tests at the sizes of the BASELINE configs (BASELINE.json / BASELINE.md section 3), through the C ABI:

  cfg1  bundled `counts`, stress sub-case: all 18 801 genes fitted, 15 checked
  cfg2  synthetic  5 000 x  50 (seed 20252): dense lp/grad vs the oracle, NUTS decisions vs the oracle
  cfg3  synthetic 20 000 x 200 (seed 20253): dense lp/grad vs the oracle (automatic and forced launch geometries)
  cfg4  synthetic 50 000 x 500 (seed 20254): dense lp/grad vs the oracle, whole and gene-sharded, + the PPC kernel
  cfg5  two-pass identify_outliers(): reduced problem vs an oracle-driven two-pass, full 20 000 x 200 vs the generator
plus the committed golden vectors, the approximated analysis against its oracle restatement, and predictive draw
counts beyond what fits LDS. Tolerances as in test_gpu_parity.py: lp 1e-11 relative, gradient 1e-10 relative to
(1 + |g|), predictive integers bit-identical, intervals 1e-9.
"""
import math
import os

import numpy as np
import pytest

from oracle import independent as ind

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {"cfg2": (5000, 50, 20252), "cfg3": (20000, 200, 20253), "cfg4": (50000, 500, 20254)}


@pytest.fixture(scope="module")
def L():
    from ppcseq_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: the product has no CPU fallback")
    return _lib


def _points(d, G, K, D, seed):
    """Two evaluation points: near the generator's truth (what the sampler visits) and a rough one (initialisation-like)."""
    rng = np.random.default_rng(seed)
    u = rng.uniform(-0.3, 0.3, (2, D))
    u[0, 3:3 + G] = d["truth"]["intercept"] + rng.normal(0, 0.05, G)
    u[0, 3 + G:3 + G + K] = d["truth"]["alpha"][1, :K]
    u[0, 3 + G + K:3 + G + K + G] = d["truth"]["sigma_raw"] + rng.normal(0, 0.05, G)
    u[1] = rng.uniform(-1, 1, D)
    u[1, 3:3 + G] += 5
    return u


def _assert_lp_grad(lp, g, lpo, go, what, noise=None):
    """noise: the fp64 rounding floor of the ORACLE's gradient per coordinate. It sums y - (y + phi) mu / (mu + phi) cell
    by cell: terms of the size of the counts that cancel to a gradient near 0 at the posterior mode, i.e. an absolute error
    of a few eps * sum_s y (1e-9 for a gene with 2e7 reads); the product's phi (sum rho - n) does not cancel (the golden
    vectors, computed with mpmath, show it closer to the truth than the oracle there)."""
    assert abs(lp - lpo) <= 1e-11 * max(1.0, abs(lpo)), (what, lp, lpo)
    tol = 1e-10 * (1 + np.abs(go)) + (0.0 if noise is None else noise)
    assert np.all(np.abs(g - go) <= tol), (what, float(np.max(np.abs(g - go) / tol)))


def _oracle_noise(counts, G, K, D):
    sy = counts.sum(1).astype(np.float64)
    n = np.zeros(D)
    n[3:3 + G] = sy
    n[3 + G:3 + G + K] = sy[:K]
    n[D - 3 - G:D - 3] = sy
    return 4 * np.finfo(np.float64).eps * n


def test_golden_vectors(L):
    """Committed vectors (tests/golden/lpgrad_small.npz; generator and cross-checks in make_lpgrad_fixture.py)."""
    z = np.load(os.path.join(HERE, "golden", "lpgrad_small.npz"))
    for n in range(int(z["n_cases"])):
        g = {k: z[f"c{n}_{k}"] for k in ("counts", "X", "exposure", "K", "excl", "u", "lp", "grad")}
        m = L.Model(g["counts"], g["X"], g["exposure"], int(g["K"]), excl=g["excl"])
        try:
            lp, grad = m.log_prob_grad(g["u"])
        finally:
            m.close()
        for i in range(len(lp)):
            _assert_lp_grad(lp[i], grad[i], g["lp"][i], g["grad"][i], (n, i))


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4"])
def test_density_and_gradient_at_baseline_size(L, oracle, name):
    """Dense comparison of lp and the whole gradient with the oracle at the full size of the config."""
    G, S, seed = CONFIGS[name]
    d = ind.synth(G, S, seed=seed)
    K = d["K"]
    mo = oracle.model(d["counts"], d["X"], d["exposure"], K, n_threads=min(16, os.cpu_count() or 1))
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        u = _points(d, G, K, m.D, seed)
        noise = _oracle_noise(d["counts"], G, K, m.D)
        ref = [oracle.log_prob_grad(mo, u[i]) for i in range(2)]
        geometries = [(0, 0)] if name != "cfg3" else [(0, 0), (8, 0), (4, 0), (16, 300), (1, 0)]
        for lanes, gpw in geometries:
            m.set_launch(lanes, gpw)
            lp, g = m.log_prob_grad(u)
            for i in range(2):
                _assert_lp_grad(lp[i], g[i], ref[i][0], ref[i][1], (name, lanes, gpw, i), noise)
        if name == "cfg3":                       # 8 chains per launch, as the bench runs
            m.set_launch(0, 0)
            lp8, g8 = m.log_prob_grad(np.repeat(u[:1], 8, axis=0))
            for i in range(8):
                _assert_lp_grad(lp8[i], g8[i], ref[0][0], ref[0][1], (name, "8 points", i), noise)
        # excluded cells (pass 2 of identify_outliers) at full size
        excl = np.array([7, S + 3, (G // 2) * S + S - 1, G * S - 1], np.int32)
        m.set_exclusions(excl)
        lpx, gx = m.log_prob_grad(u[0])
        mox = oracle.model(d["counts"], d["X"], d["exposure"], K, excl=excl, n_threads=min(16, os.cpu_count() or 1))
        lpo, go = oracle.log_prob_grad(mox, u[0])
        _assert_lp_grad(lpx, gx, lpo, go, (name, "exclusions"), noise)
    finally:
        m.close()


def test_cfg2_nuts_decisions_follow_oracle(L, oracle):
    """cfg2 (5 000 x 50, 4 chains): same Philox streams, same algorithm => identical tree sizes, depths and divergences
    for the first iterations of warm-up (before floating-point chaos separates any two implementations)."""
    G, S, seed = CONFIGS["cfg2"]
    d = ind.synth(G, S, seed=seed)
    K = d["K"]
    mo = oracle.model(d["counts"], d["X"], d["exposure"], K, n_threads=min(16, os.cpu_count() or 1))
    r = oracle.nuts_model(mo, oracle.cfg(chains=4, iter=12, warmup=12, seed=20252))
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        f = m.fit_nuts(chains=4, iter=12, warmup=12, seed=20252)
        dg = f.diagnostics()
        f.close()
    finally:
        m.close()
    assert np.array_equal(dg["n_leapfrog"], r.n_leapfrog)
    assert np.array_equal(dg["treedepth"], r.treedepth)
    assert np.array_equal(dg["divergent"], r.divergent)
    assert np.max(np.abs(dg["stepsize"] - r.stepsize) / r.stepsize) < 1e-8
    assert np.max(np.abs(dg["accept"] - r.accept)) < 1e-6
    assert r.treedepth.max() >= 4 and r.n_leapfrog.sum() > 300


def test_cfg3_nuts_decisions_follow_oracle(L, oracle):
    """The headline configuration (20 000 x 200): the first warm-up iterations of two chains against the oracle's NUTS at the
    same seed -- identical tree sizes and depths, step sizes to 1e-8 (the oracle costs 0.15 s per gradient here, so the
    comparison stops after a few hundred of them; cfg2 above goes further)."""
    G, S, seed = CONFIGS["cfg3"]
    d = ind.synth(G, S, seed=seed)
    K = d["K"]
    mo = oracle.model(d["counts"], d["X"], d["exposure"], K, n_threads=min(16, os.cpu_count() or 1))
    kw = dict(chains=2, iter=10, warmup=10, seed=20253, max_treedepth=5)
    r = oracle.nuts_model(mo, oracle.cfg(**kw))
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        f = m.fit_nuts(**kw)
        dg = f.diagnostics()
        f.close()
    finally:
        m.close()
    assert np.array_equal(dg["n_leapfrog"], r.n_leapfrog)
    assert np.array_equal(dg["treedepth"], r.treedepth)
    assert np.array_equal(dg["divergent"], r.divergent)
    assert np.max(np.abs(dg["stepsize"] - r.stepsize) / r.stepsize) < 1e-8
    assert r.n_leapfrog.sum() > 80 and r.treedepth.max() >= 4


def test_cfg4_gene_shards_and_ppc_kernel(L, oracle):
    """cfg4 (50 000 x 500): the gene-sharded run (in-process shards: the exchange step is the same sum) takes the same
    decisions as the unsharded run, and the posterior-predictive kernel reproduces the oracle's integers on the draws."""
    G, S, seed = CONFIGS["cfg4"]
    d = ind.synth(G, S, seed=seed)
    K = d["K"]
    kw = dict(chains=2, iter=6, warmup=4, seed=11, max_treedepth=4)
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        f = m.fit_nuts(**kw)
        dg, dr = f.diagnostics(), f.draws()
        # PPC on a slice of the checked genes is checked below through a small model holding the same parameters
        ci = f.ppc(0.7352941, 0.01, 0.99, seed=3)
        f.close()
    finally:
        m.close()
    bounds = [G * k // 4 for k in range(5)]
    shards = [L.Model(d["counts"][a:b], d["X"], d["exposure"], 0, shard=(G, K, a, b)) for a, b in zip(bounds[:-1], bounds[1:])]
    try:
        fits = L.fit_nuts_shards(shards, **kw)
        dgs = fits[0].diagnostics()
        lo = [fi.draws() for fi in fits]
        for fi in fits:
            fi.close()
    finally:
        for s_ in shards:
            s_.close()
    assert np.array_equal(dgs["n_leapfrog"], dg["n_leapfrog"]) and np.array_equal(dgs["treedepth"], dg["treedepth"])
    # shard 0 holds hyper-parameters + its genes: its intercept columns are the first G/4 of the whole run
    assert np.max(np.abs(lo[0][..., 3:3 + bounds[1]] - dr[..., 3:3 + bounds[1]])) < 1e-6
    # predictive summary of the whole run against the oracle on the same draws (first 40 checked genes: the oracle's cost)
    Ks = 40
    mo = oracle.model(d["counts"], d["X"], d["exposure"], K, n_threads=min(16, os.cpu_count() or 1))
    gq = oracle.generated_quantities(mo, dr.reshape(-1, dr.shape[-1]), 0.7352941, seed=3)[:, :Ks]
    assert np.max(np.abs(oracle.summarise(gq, 0.01, 0.99) - ci[:Ks])) < 1e-9
    assert np.isfinite(ci).all()


def test_approximated_analysis_matches_its_oracle_restatement(L, oracle):
    """fit_to_counts_rng_approximated (R/utilities.R:733-784): resampled posterior, n_gen predictive draws per cell.
    Same Philox specification on both sides => bit-identical integers and 1e-9 intervals, also with the truncation
    compensation of the test pass and more draws than posterior samples."""
    d = ind.synth(30, 8, K=4, seed=3)
    mo = oracle.model(d["counts"], d["X"], d["exposure"], 4)
    m = L.Model(d["counts"], d["X"], d["exposure"], 4)
    try:
        f = m.fit_nuts(chains=3, iter=250, warmup=150, seed=2)
        dr = f.draws().reshape(-1, f.D)
        for tc, p, n_gen, sd in [(1.0, 0.05, 1000, 5), (0.7352941, 0.002, 5000, 8), (0.7352941, 0.01, 137, 9)]:
            ci, rng = f.ppc(tc, p, 1 - p, seed=sd, n_gen=n_gen, resample=True, return_counts_rng=True)
            gq = oracle.generated_quantities_approx(mo, dr, n_gen, tc, seed=sd)
            assert np.array_equal(gq, rng)
            assert np.max(np.abs(oracle.summarise(gq, p, 1 - p) - ci)) < 1e-9
        f.close()
    finally:
        m.close()


def test_more_predictive_draws_than_fit_in_lds(L, oracle):
    """how_many_posterior_draws = draws_after_tail / threshold reaches 100 000 at the reference's defaults with 200
    samples (R/methods.R:166-167): beyond 39 680 draws per cell the kernel keeps them in a global scratch buffer. Also the
    hand-over from the wavefront-per-cell kernel (up to 4096 draws per cell) to the workgroup-per-cell kernel."""
    d = ind.synth(12, 5, K=2, seed=5)
    mo = oracle.model(d["counts"], d["X"], d["exposure"], 2, n_threads=8)
    m = L.Model(d["counts"], d["X"], d["exposure"], 2)
    try:
        f = m.fit_nuts(chains=3, iter=250, warmup=150, seed=4)
        dr = f.draws().reshape(-1, f.D)
        for n_gen in (4095, 4096, 4097, 39680, 39681, 100000):
            ci, rng = f.ppc(0.7352941, 0.0005, 0.9995, seed=6, n_gen=n_gen, resample=True, return_counts_rng=True)
            gq = oracle.generated_quantities_approx(mo, dr, n_gen, 0.7352941, seed=6)
            assert np.array_equal(gq, rng), n_gen
            ref = oracle.summarise(gq, 0.0005, 0.9995)
            assert np.array_equal(ref[..., 2:], ci[..., 2:]), n_gen                       # quantiles of identical integers: exact
            assert np.max(np.abs(ref[..., :2] - ci[..., :2]) / (1 + np.abs(ref[..., :2]))) < 1e-11, n_gen   # mean, sd: summation order
        f.close()
    finally:
        m.close()
    # many cells per workgroup of the scratch path: more cells than the 1024 workgroups it launches
    d2 = ind.synth(300, 8, K=140, seed=6)
    m2 = L.Model(d2["counts"], d2["X"], d2["exposure"], 140)
    mo2 = oracle.model(d2["counts"], d2["X"], d2["exposure"], 140, n_threads=8)
    try:
        f2 = m2.fit_nuts(chains=3, iter=170, warmup=150, seed=4)
        dr2 = f2.draws().reshape(-1, f2.D)
        ci2 = f2.ppc(1.0, 0.01, 0.99, seed=7, n_gen=40000, resample=True)
        f2.close()
    finally:
        m2.close()
    gq2 = oracle.generated_quantities_approx(mo2, dr2, 40000, 1.0, seed=7)[:, :3]
    ref2 = oracle.summarise(gq2, 0.01, 0.99)
    assert np.array_equal(ref2[..., 2:], ci2[:3, :, 2:]) and np.max(np.abs(ref2 - ci2[:3]) / (1 + np.abs(ref2))) < 1e-11


def _oracle_do_inference(oracle, counts, X, expo, K, p, draws, seed, excl=None, tc=1.0, approx=False):
    """do_inference() (R/utilities.R:1321-1547) driven by the oracle: same chain / iteration arithmetic as the mirror."""
    from ppcseq_amd.inference import _post_process, find_optimal_number_of_chains
    practical = 1000 if approx else draws
    chains = max(3, min(8, find_optimal_number_of_chains(practical)))
    n_iter = int(math.ceil(practical / chains)) + 150
    mo = oracle.model(counts, X, expo, K, excl=excl, n_threads=min(16, os.cpu_count() or 1))
    r = oracle.nuts_model(mo, oracle.cfg(chains=chains, iter=n_iter, warmup=150, seed=seed))
    dr = r.draws.reshape(-1, r.draws.shape[-1])
    if approx:
        gq = oracle.generated_quantities_approx(mo, dr, int(draws), tc, seed=seed)
    else:
        gq = oracle.generated_quantities(mo, dr, tc, seed=seed)
    ci = oracle.summarise(gq, p, 1 - p)
    off = 3 + counts.shape[0]
    return _post_process(counts[:K], ci, dr[:, off:off + K].mean(0), X), chains, n_iter


def test_cfg5_two_pass_reduced_against_oracle_two_pass(L, oracle):
    """cfg5 on a reduced problem (300 genes x 24 samples, 20 checked, pfp = 5): identify_outliers() on the GPU against
    the same two passes driven by the oracle at the same seed (R/methods.R:268-342): discovery flags, the excluded set fed
    back, test-pass flags. The two samplers are independent realisations after the first iterations (floating-point
    chaos), so a flag may differ only where the count sits within the Monte-Carlo spread of the interval end."""
    from ppcseq_amd.inference import do_inference
    d = ind.synth(300, 24, K=20, seed=505)
    counts, X, expo, K = d["counts"], d["X"], d["exposure"], d["K"]
    S = counts.shape[1]
    thr2 = 5 / 100 / S * 2
    thr1 = max(0.05, 2 * thr2)
    draws1, draws2 = max(1000, 10 / thr1), max(1000, 10 / thr2)
    seed = 77

    def margin(y, ci_lo, ci_hi):
        return np.minimum(np.abs(y - ci_hi) / (1 + ci_hi), np.abs(y - ci_lo) / (1 + ci_lo))

    g1 = do_inference(counts, X, expo, K, cores=8, adj_prob_theshold=thr1, how_many_posterior_draws=draws1, seed=seed)
    o1, ch, it = _oracle_do_inference(oracle, counts, X, expo, K, thr1, draws1, seed)
    assert (g1.chains, g1.iter) == (ch, it)
    y = counts[:K].astype(float)
    diff1 = g1.deleterious_outliers != o1.deleterious_outliers
    assert diff1.mean() < 0.03 and np.all(margin(y, o1.lower, o1.upper)[diff1] < 0.35)
    # the test pass, each path with ITS OWN exclusions (what identify_outliers does), truncation compensation 0.7352941
    ex_g = np.flatnonzero(g1.deleterious_outliers.ravel()).astype(np.int32)
    ex_o = np.flatnonzero(o1.deleterious_outliers.ravel()).astype(np.int32)
    g2 = do_inference(counts, X, expo, K, cores=8, adj_prob_theshold=thr2, how_many_posterior_draws=draws2, seed=seed,
                      to_exclude=ex_g, truncation_compensation=0.7352941)
    o2, ch2, it2 = _oracle_do_inference(oracle, counts, X, expo, K, thr2, draws2, seed, excl=ex_o, tc=0.7352941)
    assert (g2.chains, g2.iter) == (ch2, it2)
    diff2 = g2.deleterious_outliers != o2.deleterious_outliers
    assert diff2.mean() < 0.03 and np.all(margin(y, o2.lower, o2.upper)[diff2] < 0.35)
    rel_up = np.abs(g2.upper - o2.upper) / (1 + o2.upper)
    assert np.median(rel_up) < 0.1
    # the gross outliers the generator injected are called by both paths
    for (gi, si) in d["injected"]:
        assert g2.deleterious_outliers[gi, si] == o2.deleterious_outliers[gi, si]
    assert sum(bool(g2.deleterious_outliers[gi, si]) for gi, si in d["injected"]) >= 1


def _tidy(counts, X, K, rng):
    import pandas as pd
    G, S = counts.shape
    genes = np.array([f"g{i:05d}" for i in range(G)])
    samples = np.array([f"s{j:03d}" for j in range(S)])
    gi, sj = np.meshgrid(np.arange(G), np.arange(S), indexing="ij")
    pval = np.concatenate([np.full(K, 1e-6), rng.uniform(0.01, 1, G - K)])
    return genes, pd.DataFrame({"symbol": genes[gi.ravel()], "sample": samples[sj.ravel()], "value": counts.ravel().astype(np.int64),
                                "Label": np.where(X[sj.ravel(), 1] > 0.5, "B", "A"), "PValue": pval[gi.ravel()],
                                "is_significant": (gi.ravel() < K)})


def test_cfg5_two_pass_full_size_against_generator_truth(L):
    """cfg5 as named: the cfg3 matrix (20 000 x 200) through identify_outliers() -- discovery pass, exclusion, test pass
    with 20 000 predictive draws per cell and truncation compensation, pfp = 5, K = 1000. The CPU path cannot run this
    size (hours per fit), so the check is against the generator: the injected outliers come back as deleterious outliers
    and the clean checked genes are called at about the nominal rate."""
    from ppcseq_amd.methods import identify_outliers
    G, S, seed = CONFIGS["cfg3"]
    d = ind.synth(G, S, seed=seed)
    K = d["K"]
    genes, df = _tidy(d["counts"], d["X"], K, np.random.default_rng(1))
    res = identify_outliers(df, formula="~ Label", sample="sample", transcript="symbol", abundance="value",
                            significance="PValue", do_check="is_significant", percent_false_positive_genes=5,
                            how_many_negative_controls=G - K, seed=20255, cores=8,
                            approximate_posterior_inference=False, approximate_posterior_analysis=False)
    assert len(res) == K and res.attrs["total_draws"] == S * K * 20000
    hit = 0
    for g, s in d["injected"]:
        sw = res.loc[res["symbol"] == genes[g], "sample_wise_data"].iloc[0]
        hit += bool(sw["deleterious_outliers"].to_numpy()[s])
    assert hit >= 0.78 * len(d["injected"])                # 82-83 of 100 at five sampler seeds (profiles/r03_cfg5_seeds.json)
    inj = {g for g, _ in d["injected"]}
    flag = dict(zip(res["symbol"], res["tot_deleterious_outliers"]))
    clean = [i for i in range(K) if i not in inj]
    # nominal 5 % of genes; 6.2-7.3 % at five sampler seeds, mostly the same genes: the data's and the threshold arithmetic's
    # property (profiles/r03_cfg5_seeds.json), not sampler noise
    assert sum(flag[genes[i]] > 0 for i in clean) / len(clean) < 0.09


def test_cfg1_stress_all_genes_as_controls(L, bundled):
    """cfg1 stress sub-case: the bundled counts with ALL 18 801 genes in the fit, the README's 15 checked
    genes (FDR < 0.01), ~ Label, pfp = 5. The README's calls -- CYP1A1 and LYZ, one deleterious outlier each
    (README.md:75-92) -- must survive the 36-fold larger set of controls."""
    import pandas as pd
    from ppcseq_amd.methods import identify_outliers
    genes = [str(g) for g in bundled["genes"]]
    samples = [str(s) for s in bundled["samples"]]
    G, S = len(genes), len(samples)
    df = pd.DataFrame({
        "symbol": np.repeat(genes, S), "sample": np.tile(samples, G), "value": bundled["value"].reshape(-1),
        "PValue": np.repeat(bundled["PValue"], S), "Label": np.tile(bundled["Label"].astype(str), G)})
    df["is_significant"] = np.repeat(bundled["FDR"] < 0.01, S)
    assert df["is_significant"].sum() == 15 * S
    res = identify_outliers(df, formula="~ Label", sample="sample", transcript="symbol", abundance="value",
                            significance="PValue", do_check="is_significant", percent_false_positive_genes=5,
                            how_many_negative_controls=G - 15, seed=20251, cores=3,
                            approximate_posterior_inference=False, approximate_posterior_analysis=False)
    called = set(res.loc[res["tot_deleterious_outliers"] > 0, "symbol"])
    assert {"CYP1A1", "LYZ"} <= called and len(called) <= 4
    assert res.attrs["diagnostics_test"]["divergent"][:, 150:].mean() < 0.02


def test_cfg3_warmup_follows_the_oracle_at_full_tree_depth(L):
    """BASELINE config 3 at full size against the oracle's NUTS at Stan's defaults (max_treedepth 10): the first 45 warm-up
    iterations of 2 chains. The oracle needs ~0.25 s per gradient here, so its run is a committed fixture
    (tests/golden/cfg3_nuts_oracle.npz, written by tests/golden/make_cfg3_nuts_fixture.py: 2144 gradient evaluations, trees up to
    255 leapfrogs). Tree sizes, depths and divergences must be identical in every one of the 45 iterations; step sizes agree
    to 1e-8 over the first 25 and to 1e-6 over all 45 (rounding differences of the two implementations grow along the
    trajectories: 0 at the start, 1e-11 up to iteration 29, 2e-7 at iteration 45 with the 16 lanes per gene a 2-chain fit
    of this size runs with), acceptance statistics to 1e-5 -- in both round structures."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg3_nuts_oracle.npz"))
    G, S, data_seed, chains, n_iter, seed = (int(v) for v in z["config"])
    assert (G, S, data_seed) == (CONFIGS["cfg3"][0], CONFIGS["cfg3"][1], CONFIGS["cfg3"][2]) and n_iter >= 45
    d = ind.synth(G, S, seed=data_seed)
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    try:
        for pipe in (-1, 0):
            m.set_rounds(pipelined=pipe)
            f = m.fit_nuts(chains=chains, iter=n_iter, warmup=n_iter, seed=seed)
            dg = f.diagnostics()
            f.close()
            assert np.array_equal(dg["n_leapfrog"], z["n_leapfrog"]), (pipe, dg["n_leapfrog"].tolist(), z["n_leapfrog"].tolist())
            assert np.array_equal(dg["treedepth"], z["treedepth"]) and np.array_equal(dg["divergent"], z["divergent"])
            rel = np.abs(dg["stepsize"] / z["stepsize"] - 1)
            assert rel[:, :25].max() < 1e-8 and rel.max() < 1e-6, (pipe, rel.max(axis=0))
            assert np.max(np.abs(dg["accept"] - z["accept"])) < 1e-5
        assert z["n_leapfrog"].max() >= 255 and z["n_leapfrog"].sum() > 2000
    finally:
        m.close()


def test_a_chain_does_not_depend_on_the_chains_it_shares_launches_with(L, monkeypatch):
    """Chains are independent given their global id (Philox key) and the lanes per gene (summation order inside a gene).
    The same chain must therefore come out bit-identical from a 3-chain fit, from a fit with far more chains than the
    chip holds resident workgroups per chain for (130 chains: the launch plan has fewer than 8 workgroups per chain, and
    chains finish at different times, so the plan is redone many times), and from a fit whose chains are split into groups
    on separate streams (ppcx_model_set_rounds)."""
    d = ind.synth(300, 12, K=20, seed=17)
    m = L.Model(d["counts"], d["X"], d["exposure"], 20)
    try:
        m.set_launch(8, 0)
        kw = dict(iter=60, warmup=40, seed=5)
        f3 = m.fit_nuts(chains=3, **kw)
        d3, g3 = f3.draws().copy(), f3.diagnostics()["n_leapfrog"].copy()
        f3.close()
        f130 = m.fit_nuts(chains=130, **kw)
        d130, g130 = f130.draws(), f130.diagnostics()["n_leapfrog"]
        assert np.array_equal(g130[:3], g3) and np.array_equal(d130[:3], d3)
        assert len({tuple(r) for r in g130.tolist()}) > 100          # the chains are genuinely different trajectories
        f130.close()
        m.set_rounds(stream_groups=3)
        f7 = m.fit_nuts(chains=7, **kw)
        assert np.array_equal(f7.diagnostics()["n_leapfrog"][:3], g3) and np.array_equal(f7.draws()[:3], d3)
        f7.close()
        # a chain placed by its global id: chain 2 of the 3-chain fit alone in a fit with chain_id_offset = 2
        m.set_rounds(stream_groups=0)
        f1 = m.fit_nuts(chains=1, chain_id_offset=2, **kw)
        assert np.array_equal(f1.draws()[0], d3[2])
        f1.close()
    finally:
        m.close()


def test_cfg3_chain_groups_run_the_single_stream_chains(L):
    """At BASELINE size: a fit in chain groups -- whose launches are trimmed to whole passes per wavefront and share the chip with
    the other groups'  -- gives every chain the draws it has on one stream, bit for bit."""
    G, S = CONFIGS["cfg3"][0], CONFIGS["cfg3"][1]
    d = ind.synth(G, S, seed=CONFIGS["cfg3"][2])
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    try:
        m.set_launch(8, 0)                                               # the lanes per gene of an 8-chain fit, for every fit below
        kw = dict(chains=5, iter=45, warmup=30, seed=11)
        m.set_rounds(stream_groups=1)
        f = m.fit_nuts(**kw); lp1, n1 = f.diagnostics()["lp"].copy(), f.diagnostics()["n_leapfrog"].copy(); f.close()
        for groups in (2, 3):
            m.set_rounds(stream_groups=groups)
            assert m.get_rounds(5) == (True, groups)
            f = m.fit_nuts(**kw)
            assert np.array_equal(f.diagnostics()["n_leapfrog"], n1) and np.array_equal(f.diagnostics()["lp"], lp1), groups
            f.close()
        lanes, nb, _ = m.get_plan(2)
        lanes_t, nb_t, _ = m.get_plan(-2)
        assert lanes_t == 8 and nb_t < nb                                # the groups' launches really are the trimmed ones here
    finally:
        m.close()


@pytest.mark.parametrize("G,S,K", [(20000, 200, 1000), (5000, 50, 250), (300, 12, 20), (7, 3, 2)])
def test_launch_plan_invariants(L, G, S, K):
    """The host's plan of a log-likelihood launch: the wavefronts' ranges tile the gene order and none holds more passes
    (of 64 / L genes, counted from its own start) than its share rounded up -- for every number of chains still running."""
    d = ind.synth(G, S, K=K, seed=3)
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        for nch in (1, 2, 3, 5, 8, 16, 130):
            lanes, nb, b = m.get_plan(nch)
            gpw = 64 // lanes
            npass = -(-G // gpw)
            wpc = 4 * nb
            assert b[0] == 0 and b[-1] == G and np.all(np.diff(b) >= 0)
            pmax = -(-npass // wpc)
            assert np.max(-(-np.diff(b) // gpw)) <= pmax, (nch, lanes, nb)
            assert nb * nch <= max(4 * 256, nch)               # resident workgroups of the chip (4 per CU), or one per chain
            # the launch of one of several chain groups: not more workgroups than it takes to give every wavefront the passes
            # of the busiest one -- the same number of passes at most, the same tiling, the slots left to the other groups
            lanes_t, nb_t, bt = m.get_plan(-nch)
            assert lanes_t == lanes and nb_t <= nb
            assert bt[0] == 0 and bt[-1] == G and np.all(np.diff(bt) >= 0)
            assert -(-npass // (4 * nb_t)) == pmax and np.max(-(-np.diff(bt) // gpw)) <= pmax, (nch, lanes, nb, nb_t)
            assert nb_t == nb or 4 * (nb_t - 8) * pmax < npass        # trimmed to the fewest (whole runs of 8 workgroups)
    finally:
        m.close()


def test_lanes_per_gene_are_chosen_for_the_chains_of_a_launch(L):
    """The automatic lanes per gene at cfg3 size for launches of 1..8 chains (round 5: passes of the busiest wavefront x (8 + cell
    iterations of a lane), ppcx_capi.hip choose_launch): none below 4 lanes (whose requests touch 16+ times the cache lines), and the
    choices measured fastest per number of chains (kernel level: 1 chain L = 8, 2 and 3 chains L = 4, 8 chains L = 8). A fit chooses
    for the chains of ONE launch -- its chain groups' (three groups from eight chains on) -- not for all of its chains."""
    G, S = CONFIGS["cfg3"][0], CONFIGS["cfg3"][1]
    d = ind.synth(G, S, seed=CONFIGS["cfg3"][2])
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    try:
        chosen = []
        for nch in range(1, 9):
            lanes, nb, b = m.get_plan(nch)
            chosen.append(lanes)
            assert lanes >= 4 and nb * nch <= 4 * 256, (nch, lanes, nb)
        assert chosen == [8, 4, 4, 8, 4, 4, 8, 8], chosen
        kw = dict(iter=3, warmup=2, seed=1)
        m.fit_nuts(chains=8, **kw).close()                               # groups of 3, 3 and 2 chains
        assert m.get_launch()[0] == 4
        m.fit_nuts(chains=1, **kw).close()
        assert m.get_launch()[0] == 8
        m.set_rounds(stream_groups=1)                                    # all eight chains in every launch
        m.fit_nuts(chains=8, **kw).close()
        assert m.get_launch()[0] == 8
    finally:
        m.close()


@pytest.mark.parametrize("G,S,C,K", [(24, 2500, 2, 6), (16, 4300, 2, 0), (9, 1, 1, 0), (5, 3, 2, 5)])
def test_extreme_sample_counts(L, oracle, G, S, C, K):
    """Many samples: the per-sample constants need more than the default 64 KB of dynamic LDS (88 KB at S = 2500, 146 KB at
    S = 4300: one resident workgroup per CU, the launch plan follows); and degenerate ones (a single sample; fewer samples
    than lanes per gene). Density and gradient against the oracle."""
    d = ind.synth(G, S, K=K, seed=8, C=C)
    mo = oracle.model(d["counts"], d["X"], d["exposure"], K)
    m = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        rng = np.random.default_rng(1)
        u = rng.uniform(-0.5, 0.5, (3, m.D))
        u[:, 3:3 + G] += 4.0
        for lanes in (0, 8, 64):
            m.set_launch(lanes, 0)
            lp, g = m.log_prob_grad(u)
            for i in range(3):
                lpo, go = oracle.log_prob_grad(mo, u[i])
                assert abs(lp[i] - lpo) <= 1e-11 * max(1.0, abs(lpo)), (lanes, i)
                assert np.max(np.abs(g[i] - go) / (1 + np.abs(go))) <= 1e-10, (lanes, i)
    finally:
        m.close()


def test_too_many_samples_for_lds_is_refused(L):
    """S * C doubles of per-sample constants beside 20 KB of tables (round 5; S * (2 + C) before): 5 300 samples of a two-group
    design now run (tests/test_gpu_parity.py goes to 8 000), 9 500 do not fit the 160 KB and are refused with a status."""
    d = ind.synth(4, 5300, K=0, seed=2)
    L.Model(d["counts"], d["X"], d["exposure"], 0).close()
    d = ind.synth(4, 9500, K=0, seed=2)
    with pytest.raises(L.PpcxError, match="LDS"):
        L.Model(d["counts"], d["X"], d["exposure"], 0)


def test_fully_excluded_gene_low_count_exclusions_and_all_genes_checked(L, oracle):
    """Exclusions that empty a whole gene, that hit cells of the low-count list (their tallies must follow), with every gene
    checked (K = G: all genes carry a slope) -- density and gradient against the oracle, then back to no exclusions."""
    rng = np.random.default_rng(11)
    G, S = 30, 14
    d = ind.synth(G, S, K=G, seed=12)
    counts = d["counts"].copy()
    counts[3] = rng.integers(0, 8, S)                     # a gene made of list cells only
    counts[4, :5] = [0, 7, 8, 1, 0]
    excl = np.concatenate([np.arange(7 * S, 8 * S),        # gene 7: every cell excluded
                           3 * S + np.array([0, 5, 9]),    # list cells of gene 3
                           4 * S + np.array([1, 2])]).astype(np.int32)
    u = rng.uniform(-0.7, 0.7, (2, oracle.dim(G, 2, G)))
    u[:, 3:3 + G] += 3.5
    m = L.Model(counts, d["X"], d["exposure"], G)
    try:
        for ex in (excl, None, excl[::-1].copy()):
            m.set_exclusions(ex)
            mo = oracle.model(counts, d["X"], d["exposure"], G, excl=ex)
            lp, g = m.log_prob_grad(u)
            for i in range(2):
                lpo, go = oracle.log_prob_grad(mo, u[i])
                assert abs(lp[i] - lpo) <= 1e-11 * max(1.0, abs(lpo))
                assert np.max(np.abs(g[i] - go) / (1 + np.abs(go))) <= 1e-10
    finally:
        m.close()
