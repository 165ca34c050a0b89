"""Host-side mirrors of the reference's R helpers (no GPU): chain chooser, type-7 quantiles, thresholds,
formula / design matrix, flags, TMM, ESS, .rda reader."""
import numpy as np
import pytest

from ppcseq_amd import inference as inf
from ppcseq_amd import methods as meth
from ppcseq_amd.ess import ess_bulk, rhat


def test_find_optimal_number_of_chains():
    # R/utilities.R:291-303; SURVEY App. C: 1000 draws -> 3 chains, 10 500 -> 8
    assert inf.find_optimal_number_of_chains(1000) == 3
    assert inf.find_optimal_number_of_chains(10500) == 8


def test_quantile_type7_matches_numpy_linear():
    x = np.random.default_rng(0).integers(0, 1000, 997)
    for p in [0.0, 0.001, 0.025, 0.5, 0.975, 1.0]:
        assert inf.quantile7(x, p) == pytest.approx(np.quantile(x, p, method="linear"), rel=1e-14)


def test_parse_formula_and_design_matrix():
    import pandas as pd
    assert meth.parse_formula("~ Label") == ["Label"]
    assert meth.parse_formula("~ 1") == []
    assert meth.parse_formula("~ a + b") == ["a", "b"]
    with pytest.raises(ValueError):
        meth.parse_formula("y ~ a")
    df = pd.DataFrame({"sample": ["s2", "s1", "s3", "s2"], "Label": ["B", "A", "B", "B"], "x": [1.0, 2.0, 3.0, 1.0]})
    X, names, samples = meth.create_design_matrix(df, "~ Label + x", "sample")
    assert samples == ["s1", "s2", "s3"] and names == ["(Intercept)", "LabelB", "x"]
    assert np.array_equal(X, np.array([[1, 0, 2.0], [1, 1, 1.0], [1, 1, 3.0]]))


def test_flags_follow_reference_rules():
    # check_if_within_posterior (R/utilities.R:651-663) and add_deleterious_if_covariate_exists (:493-513)
    counts = np.array([[10, 100, 5, 50]])
    ci = np.zeros((1, 4, 4))
    ci[0, :, 0] = [20, 20, 20, 20]        # mean
    ci[0, :, 2] = [10, 10, 10, 10]        # lower (inclusive)
    ci[0, :, 3] = [50, 50, 50, 50]        # upper (inclusive)
    X = np.array([[1, 0], [1, 0], [1, 1], [1, 1.0]])
    r = inf._post_process(counts, ci, np.array([0.7]), X)
    assert r.ppc.tolist() == [[True, False, False, True]]
    assert r.is_higher_than_mean.tolist() == [[False, True, False, False]]
    assert r.is_group_high.tolist() == [[False, False, True, True]]
    # outlier too high in the LOW group and too low in the HIGH group both work against a positive slope
    assert r.deleterious_outliers.tolist() == [[False, False, False, False]]
    r2 = inf._post_process(counts, ci, np.array([-0.7]), X)
    assert r2.deleterious_outliers.tolist() == [[False, True, True, False]]


def test_host_rules_equal_their_oracle_side_restatements(bundled):
    """The product's flag rules, chain arithmetic and TMM exposures (ppcseq_amd.inference / ppcseq_amd.methods: array
    expressions) against the oracle side's own restatements of the same R statements (oracle/independent.py: loops over cells),
    which is what pins the oracle on the reference's known answers (tests/test_oracle_reference_cases.py). Random cells --
    including counts exactly on an interval's end (dplyr::between is inclusive) and on the mean, zero and negative slopes,
    designs with and without a covariate -- and the bundled data for the exposures."""
    from oracle import independent as ind
    from ppcseq_amd import methods
    rng = np.random.default_rng(12)
    for trial in range(20):
        K, S = int(rng.integers(1, 7)), int(rng.integers(2, 12))
        lower = rng.integers(0, 40, (K, S)).astype(float)
        upper = lower + rng.integers(0, 60, (K, S))
        mean = lower + (upper - lower) * rng.random((K, S))
        counts = rng.integers(0, 120, (K, S))
        pick = rng.random((K, S))
        counts = np.where(pick < 0.15, lower, np.where(pick < 0.3, upper, np.where(pick < 0.4, np.floor(mean), counts))).astype(np.int64)
        slope = rng.normal(0, 1, K); slope[rng.random(K) < 0.2] = 0.0
        X = np.stack([np.ones(S), rng.integers(0, 2, S).astype(float)], 1) if trial % 3 else np.ones((S, 1))
        if trial % 5 == 4:
            X = np.stack([np.ones(S), rng.normal(0, 1, S)], 1)            # a continuous covariate: the split is at its mean
        ci = np.stack([mean, np.zeros((K, S)), lower, upper], -1)
        a = inf._post_process(counts, ci, slope, X)
        b = ind.flags_reference(counts, mean, lower, upper, slope, X)
        assert np.array_equal(a.ppc, b.ppc) and np.array_equal(a.is_higher_than_mean, b.is_higher_than_mean)
        if X.shape[1] > 1:
            assert np.array_equal(a.is_group_high, b.is_group_high) and np.array_equal(a.deleterious_outliers, b.deleterious_outliers)
        else:
            assert a.deleterious_outliers is None and b.deleterious_outliers is None
    for draws in (200, 1000, 1001, 4000, 10500, 100000):
        assert inf.find_optimal_number_of_chains(draws) == ind.optimal_number_of_chains(draws)
    assert ind.optimal_number_of_chains(1000) == 3 and ind.optimal_number_of_chains(10500) == 8
    from tests.conftest import bundled_test_config
    counts, _, _, _ = bundled_test_config(bundled)
    mult, nf = methods.get_scaled_counts_bulk(counts, list(range(counts.shape[1])))
    mult2, nf2 = ind.scaled_multipliers_reference(counts)
    assert np.allclose([mult[s] for s in range(counts.shape[1])], mult2, rtol=1e-12, atol=0)
    assert np.allclose([nf[s] for s in range(counts.shape[1])], nf2, rtol=1e-12, atol=0)
    m2 = rng.poisson(rng.gamma(2.0, 50.0, (400, 1)) * rng.uniform(0.5, 2.0, (1, 9))).astype(np.int64)   # zeros and ties included
    mult, nf = methods.get_scaled_counts_bulk(m2, list(range(9)))
    mult2, nf2 = ind.scaled_multipliers_reference(m2)
    assert np.allclose([mult[s] for s in range(9)], mult2, rtol=1e-12, atol=0)


def test_threshold_arithmetic_of_identify_outliers():
    # R/methods.R:156-167 with 21 samples, pfp = 1, detrimental only
    thr2 = 1 / 100 / 21 * 2
    thr1 = max(0.05, 2 * thr2)
    assert thr1 == 0.05
    assert max(10 / thr1, 1000) == 1000 and max(10 / thr2, 1000) == pytest.approx(10500)


def test_tmm_recovers_known_scaling():
    rng = np.random.default_rng(3)
    base = rng.gamma(2.0, 200.0, size=2000)
    scale = np.array([1.0, 2.0, 0.5, 1.5])
    mat = rng.poisson(base[:, None] * scale[None, :])
    mult, nf = meth.get_scaled_counts_bulk(mat, ["a", "b", "c", "d"])
    m = np.array([mult[s] for s in "abcd"])
    # multiplier brings every library to the reference sample's scale
    scaled = mat.sum(0) * m
    assert np.all(np.abs(scaled / scaled[np.argmax(np.median(mat, axis=0))] - 1) < 0.03)
    assert abs(np.exp(np.mean(np.log(list(nf.values())))) - 1) < 1e-12


def test_tmm_known_answers_from_the_published_definition():
    """edgeR is not installed (parity of the TMM step is unpinned), so the factors are checked against
    what the PUBLISHED definition gives in closed form (Robinson & Oshlack 2010: weighted mean of log-ratios after trimming
    30 % of the M values and 5 % of the A values on each side, relative to a reference sample, scaled to unit geometric
    mean):
      * composition bias: sample B = sample A with 20 % of its genes four times as abundant. With r = total_B / total_A
        (about 1.6) the unchanged genes have proportions 1/r of A's, the changed ones 4/r; trimming removes the changed ones
        and every M value that is left equals log2(1/r), so f_B / f_A = 1/r exactly;
      * the factor of a sample does not depend on its sequencing depth (TMM works on proportions);
      * identical samples get factor 1."""
    n = 1000
    a = np.round(np.geomspace(50, 50000, n)).astype(float)           # no zeros, no ties in A
    b = a.copy()
    b[::5] *= 4.0                                                     # every fifth gene: 20 %
    f = meth.tmm_norm_factors(np.stack([a, b], axis=1), ref_col=0)
    r = b.sum() / a.sum()
    assert 1.5 < r < 1.7
    assert f[1] / f[0] == pytest.approx(1 / r, rel=1e-12)
    assert f[0] * f[1] == pytest.approx(1.0, rel=1e-12)              # unit geometric mean
    assert f[0] == pytest.approx(np.sqrt(r), rel=1e-12)
    # depth invariance: sample B sequenced three times as deep
    f3 = meth.tmm_norm_factors(np.stack([a, 3.0 * b], axis=1), ref_col=0)
    assert f3 == pytest.approx(f, rel=1e-12)
    # identical samples
    assert meth.tmm_norm_factors(np.stack([a, a, 2 * a], axis=1), ref_col=0) == pytest.approx(np.ones(3), rel=1e-12)
    # the reference's multiplier on top (R/tidybulk.R:220-225): tot_ref / (tot_s * nf_s) puts B's unchanged genes on A's scale
    mult, nf = meth.get_scaled_counts_bulk(np.stack([a, b], axis=1), ["A", "B"])
    ref = "A" if np.median(a) >= np.median(b) else "B"
    scaled_a, scaled_b = a * mult["A"], b * mult["B"]
    keep = np.ones(n, bool); keep[::5] = False
    assert np.allclose(scaled_b[keep] / scaled_a[keep], 1.0, rtol=1e-12) and ref in ("A", "B")   # the bias is gone


def test_ess_estimator():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(4, 1000))
    assert 3000 < ess_bulk(x) < 5000 and abs(rhat(x) - 1) < 0.01
    y = np.zeros((4, 1000))
    e = rng.normal(size=(4, 1000))
    for t in range(1, 1000):
        y[:, t] = 0.9 * y[:, t - 1] + e[:, t]
    assert 120 < ess_bulk(y) < 400           # theory: 4000 * (1-0.9)/(1+0.9) = 210
    z = x + np.arange(4)[:, None]            # chains that disagree
    assert rhat(z) > 1.5


def test_bundled_fixture_matches_reference_facts(bundled):
    # man/counts.Rd:8, README.md:32-45, SURVEY App. E
    assert bundled["value"].shape == (18801, 21)
    genes = [str(g) for g in bundled["genes"]]
    assert genes[:3] == ["SLC16A12", "CYP1A1", "ART3"]
    assert bundled["value"][genes.index("CYP1A1")].tolist() == [6, 12, 0, 2, 0, 2, 3, 50, 2, 48, 26, 4, 4, 4, 10, 820, 5835, 2, 12, 0, 2]
    assert str(bundled["samples"][16]) == "11165PP"
    assert int((bundled["FDR"] < 0.01).sum()) == 15 and int(bundled["value"].max()) == 2580228


def test_test_config_selection(bundled):
    from tests.conftest import bundled_test_config
    counts, X, genes, K = bundled_test_config(bundled)
    assert counts.shape == (53, 21) and K == 3 and genes[:3] == ["SLC16A12", "CYP1A1", "ART3"]
    assert genes[3:8] == ["AATF", "ABCA9", "ABT1", "BHLHE40", "C1orf174"]
    assert int(counts.sum()) == 2080390 and int(counts.max()) == 24912
    assert X[:, 1].sum() == 11                                               # 10 High / 11 Neoadjuvant


def test_identify_outliers_argument_validation():
    """Error behaviour of identify_outliers mirrors R/methods.R:108-153 (raised before any GPU work)."""
    import pandas as pd
    base = pd.DataFrame({"sample": ["a", "b"] * 2, "symbol": ["g1", "g1", "g2", "g2"], "value": [1, 2, 3, 4],
                         "PValue": [0.1, 0.1, 0.9, 0.9], "chk": [True, True, False, False]})
    kw = dict(formula="~ 1", sample="sample", transcript="symbol", abundance="value", significance="PValue", do_check="chk")
    with pytest.raises(ValueError):                      # missing column (check_columns_exist)
        meth.identify_outliers(base.drop(columns=["PValue"]), **kw)
    with pytest.raises(ValueError):                      # NA in a used column (check_if_any_NA)
        meth.identify_outliers(base.assign(PValue=[0.1, None, 0.9, 0.9]), **kw)
    with pytest.raises(ValueError):                      # percent_false_positive_genes outside 0..100
        meth.identify_outliers(base, percent_false_positive_genes=101, **kw)
    with pytest.raises(TypeError):                       # abundance must be integer (R/methods.R:146-153)
        meth.identify_outliers(base.assign(value=[1.0, 2.0, 3.0, 4.0]), **kw)
    with pytest.raises(ValueError):                      # VB + save_generated_quantities (R/methods.R:131-132)
        meth.identify_outliers(base, approximate_posterior_inference=True, save_generated_quantities=True, **kw)
    with pytest.warns(UserWarning):                      # nothing to check -> empty result (R/methods.R:117-127)
        out = meth.identify_outliers(base.assign(chk=False), **kw)
    assert len(out) == 0 and list(out.columns)[0] == "symbol"


def test_identify_outliers_signature_is_the_references():
    """Argument names and defaults of ppcseq::identify_outliers (R/methods.R:74-102): a call ported unchanged must select
    the same inference mode. (`.data` and the dotted column arguments lose their dot; `device` is the one addition.)"""
    import inspect
    sig = inspect.signature(meth.identify_outliers)
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["formula"] == "~ 1" and d["percent_false_positive_genes"] == 1 and d["how_many_negative_controls"] == 500
    assert d["approximate_posterior_inference"] is True and d["approximate_posterior_analysis"] is True      # R/methods.R:85-86
    assert d["draws_after_tail"] == 10 and d["save_generated_quantities"] is False and d["pass_fit"] is False
    assert d["tol_rel_obj"] == 0.01 and d["just_discovery"] is False and d["adj_prob_theshold_2"] is None     # :94-97
    assert d["scaling_factor"] is None and d["additional_parameters_to_save"] == ()
    ref_order = ["formula", "sample", "transcript", "abundance", "significance", "do_check", "scaling_factor",
                 "percent_false_positive_genes", "how_many_negative_controls", "approximate_posterior_inference",
                 "approximate_posterior_analysis", "draws_after_tail", "save_generated_quantities",
                 "additional_parameters_to_save", "cores", "pass_fit", "do_check_only_on_detrimental", "tol_rel_obj",
                 "just_discovery", "seed", "adj_prob_theshold_2"]
    assert list(sig.parameters)[1:1 + len(ref_order)] == ref_order


def test_device_memory_model_of_the_full_analysis():
    """What the device holds for a full posterior analysis (kept draws + sampler vectors) -- the quantity the guard that
    mirrors R/methods.R:178-195 compares with the free device memory. cfg5: 20 000 x 200, K = 1000, 20 000 draws, 8 chains."""
    need = meth.required_device_memory(20000, 2, 1000, 200, 20000, 8, False)
    D = 2 * 20000 + 1000 + 6
    assert need >= 8 * 20000 * D and need < 8 * 20000 * D * 1.1       # dominated by the draws: 6.6 GB
    assert meth.required_device_memory(20000, 2, 1000, 200, 1000, 8, True) < 2e9       # ADVI: 32 evaluation slots + 1000 draws


def test_hmc_warnings_follow_rstan_checks():
    """The two messages rstan gives after sampling (divergences, maximum tree depth), from a fit's diagnostics: warm-up
    iterations do not count, and the chains that hit the maximum depth are named."""
    from ppcseq_amd.inference import hmc_warnings
    depth = np.full((3, 10), 6); div = np.zeros((3, 10), int)
    depth[1, :4] = 10; div[2, 2] = 1                         # during warm-up only
    assert hmc_warnings({"treedepth": depth, "divergent": div}, warmup=4) == []
    depth[1, 7:] = 10; div[0, 5] = 1; div[2, 9] = 1
    msgs = hmc_warnings({"treedepth": depth, "divergent": div}, warmup=4)
    assert len(msgs) == 2 and "2 divergent transitions after warmup" in msgs[0]
    assert "3 transitions after warmup that exceeded the maximum treedepth of 10" in msgs[1] and "chains [1]" in msgs[1]
