"""Host-side mirror of ppcseq's public API `identify_outliers()` (R/methods.R:74-367) on pandas.

Orchestration only -- thresholds, two passes, exclusion feed-back, tidy output -- with the same
argument names and semantics as the reference; both inference passes run on the MI355X through
`do_inference` (ppcseq_amd/inference.py -> libppcx.so). Data-frame shaping mirrors
`format_input` (R/utilities.R:924-959), `select_to_check_and_house_keeping` (:628-649),
`create_design_matrix` (:887-900), `merge_results`/`format_results` (:539-608); TMM scaling mirrors
R/tidybulk.R:150-241 with edgeR's published TMM algorithm (edgeR is third party, not vendored:
"parity unpinned" for that step; pass `.scaling_factor` to bypass it as the reference allows,
R/methods.R:229-232).
"""
from __future__ import annotations

import warnings

import numpy as np

from .inference import do_inference


def parse_formula(formula: str):
    """R/utilities.R:220-225: covariate names of a one-sided formula `~ a + b`."""
    f = formula.strip()
    if not f.startswith("~"):
        raise ValueError('The formula must be of the kind "~ covariates" ')
    rhs = f[1:].strip()
    if rhs in ("1", ""):
        return []
    return [t.strip() for t in rhs.split("+") if t.strip() not in ("1", "")]


def create_design_matrix(df, formula, sample):
    """model.matrix(formula, distinct(sample, covariates) arranged by sample) (R/utilities.R:887-900).

    Numeric covariates enter as they are; categorical ones with treatment contrasts against the first
    level in sorted order (R's default for character columns)."""
    covs = parse_formula(formula)
    d = df[[sample] + covs].drop_duplicates().sort_values(sample, kind="stable")
    cols = [np.ones(len(d))]
    names = ["(Intercept)"]
    for c in covs:
        v = d[c]
        if np.issubdtype(v.dtype, np.number):
            cols.append(v.to_numpy(dtype=np.float64))
            names.append(c)
        else:
            levels = sorted(v.astype(str).unique())
            for lv in levels[1:]:
                cols.append((v.astype(str) == lv).to_numpy(dtype=np.float64))
                names.append(f"{c}{lv}")
    return np.stack(cols, axis=1), names, d[sample].tolist()


def tmm_norm_factors(mat, ref_col, logratio_trim=0.3, sum_trim=0.05, a_cutoff=-1e10):
    """edgeR::calcNormFactors(method="TMM") restated from Robinson & Oshlack (2010) / edgeR's documented
    defaults (doWeighting = TRUE). mat: genes x samples counts. Returns factors scaled to unit geometric mean."""
    from scipy.stats import rankdata
    mat = np.asarray(mat, dtype=np.float64)
    lib = mat.sum(axis=0)
    ref = mat[:, ref_col]
    nR = lib[ref_col]
    f = np.ones(mat.shape[1])
    for j in range(mat.shape[1]):
        obs, nO = mat[:, j], lib[j]
        with np.errstate(divide="ignore", invalid="ignore"):
            logR = np.log2((obs / nO) / (ref / nR))
            absE = (np.log2(obs / nO) + np.log2(ref / nR)) / 2
            v = (nO - obs) / nO / obs + (nR - ref) / nR / ref
        fin = np.isfinite(logR) & np.isfinite(absE) & (absE > a_cutoff)
        logR, absE, v = logR[fin], absE[fin], v[fin]
        if logR.size == 0 or np.max(np.abs(logR)) < 1e-6:
            f[j] = 1.0
            continue
        n = logR.size
        loL = np.floor(n * logratio_trim) + 1
        hiL = n + 1 - loL
        loS = np.floor(n * sum_trim) + 1
        hiS = n + 1 - loS
        rL, rS = rankdata(logR), rankdata(absE)
        keep = (rL >= loL) & (rL <= hiL) & (rS >= loS) & (rS <= hiS)
        val = np.sum(logR[keep] / v[keep]) / np.sum(1 / v[keep])
        f[j] = 2.0 ** (0.0 if not np.isfinite(val) else val)
    return f / np.exp(np.mean(np.log(f)))


def get_scaled_counts_bulk(mat, samples):
    """R/tidybulk.R:150-241: reference = sample whose median is closest to the max median;
    multiplier = tot_ref / (tot_s * nf_s). mat: genes x samples (selected genes only)."""
    med = np.median(mat, axis=0)
    ref = int(np.argmin(np.abs(med - med.max())))
    nf = tmm_norm_factors(mat, ref)
    tot = mat.sum(axis=0).astype(np.float64)
    mult = tot[ref] / (tot * nf)
    return dict(zip(samples, mult)), dict(zip(samples, nf))


def required_device_memory(G, C, K, S, how_many_posterior_draws, cores, approximate_posterior_inference):
    """Bytes the device holds for the full posterior analysis of one pass: the kept draws [chains][n_keep][D] stay
    resident for the posterior-predictive kernel, beside the sampler's per-chain vectors. The device analogue of the
    reference's RAM model (R/methods.R:184-188: `1.044e6 + draws * 3.777e-2` for MCMC, `1.554e6 + draws * 7.327e-2`
    for VB, units unstated)."""
    import math
    from .inference import find_optimal_number_of_chains
    D = 2 * G + K * max(C - 1, 1) + 6
    Dpad = (D + 31) // 32 * 32
    if approximate_posterior_inference:
        return 8 * (int(how_many_posterior_draws) * D + 32 * 67 * Dpad) + 4 * G * S
    chains = max(3, min(int(cores), find_optimal_number_of_chains(how_many_posterior_draws)))
    n_keep = int(math.ceil(how_many_posterior_draws / chains))
    return 8 * (chains * n_keep * D + chains * 67 * Dpad) + 4 * G * S


def identify_outliers(data, formula="~ 1", sample="sample", transcript="transcript", abundance="count",
                      significance="PValue", do_check="do_check", scaling_factor=None,
                      percent_false_positive_genes=1, how_many_negative_controls=500,
                      approximate_posterior_inference=True, approximate_posterior_analysis=True,
                      draws_after_tail=10, save_generated_quantities=False, additional_parameters_to_save=(),
                      cores=None, pass_fit=False, do_check_only_on_detrimental=None, tol_rel_obj=0.01,
                      just_discovery=False, seed=None, adj_prob_theshold_2=None, device=0, devices=None, launch=None, _pass=None):
    """Mirror of ppcseq::identify_outliers (R/methods.R:74-367): same arguments, same defaults.

    data is a tidy pandas DataFrame (one row per transcript x sample); column arguments are strings. As in the
    reference the defaults are `approximate_posterior_inference = True` (ADVI, R/methods.R:85) and
    `approximate_posterior_analysis = True` (:86); `tol_rel_obj` is accepted and, as in the reference, not used (the
    inference pass hard-codes 0.005, R/utilities.R:1492); `additional_parameters_to_save` is a development argument of
    the reference with nothing to add here (the fit keeps every parameter); `cores` defaults to the host's core count
    (`detect_cores()`), which only bounds the number of chains. Pass `approximate_posterior_inference = False` for NUTS,
    this engine's headline path. Returns a DataFrame with one row per checked transcript: <transcript>,
    sample_wise_data (nested DataFrame), ppc_samples_failed, tot_deleterious_outliers (when do_check_only_on_detrimental).

    `devices` = [...]: several HIP devices of this process. The chains of BOTH passes of a NUTS run (approximate_posterior_
    inference = False) are dealt to them, a host thread each, as the reference's sampling(chains, cores) deals its chains to
    `cores` workers in both of its passes (R/utilities.R:1500-1501, R/methods.R:268-342); the credible intervals come from the
    pooled chains. Not with save_generated_quantities / pass_fit (the draws then live on several devices). One process per GPU
    over torch.distributed: ppcseq_amd.distributed.identify_outliers. `launch` = (lanes_per_gene, workgroups) pins the
    log-likelihood launch of both passes (results are bit-identical across device counts only at equal lanes per gene).
    """
    import os
    import pandas as pd
    if cores is None:
        cores = os.cpu_count() or 1                                        # detect_cores(), R/methods.R:91
    del tol_rel_obj, additional_parameters_to_save                         # accepted for signature parity; see above
    covs = parse_formula(formula)
    if do_check_only_on_detrimental is None:
        do_check_only_on_detrimental = len(covs) > 0                      # R/methods.R:93
    for c in [sample, transcript, abundance, significance] + covs:       # check_columns_exist / check_if_any_NA
        if c not in data.columns:
            raise ValueError(f"The column {c} is not present in the data frame")
        if data[c].isna().any():
            raise ValueError(f"There are NA values in the column {c}")
    checked_rows = data[data[do_check].astype(bool)]
    if len(checked_rows) == 0:                                             # R/methods.R:117-127
        warnings.warn("ppcseq says: There are not transcripts with the category .to_check. NULL is returned.")
        return pd.DataFrame({transcript: [], "sample_wise_data": [], "ppc samples failed": [],
                             "tot deleterious_outliers": []})
    if approximate_posterior_inference and save_generated_quantities:
        raise ValueError("Variational Bayes does not support tidybayes needed for save_generated_quantities, use sampling")
    pfp = percent_false_positive_genes
    if pfp is None or not (0 <= pfp <= 100):
        raise ValueError("percent_false_positive_genes must be a string from > 0% to < 100%")
    if not np.issubdtype(data[abundance].dtype, np.integer):
        raise TypeError(f"The column {abundance} must be of class integer.")
    if seed is None:
        seed = int(np.random.default_rng().integers(1, 999999))

    n_samples = data[sample].nunique()
    if adj_prob_theshold_2 is None:                                        # R/methods.R:156-160
        adj_prob_theshold_2 = pfp / 100 / n_samples * (2 if do_check_only_on_detrimental else 1)
    adj_prob_theshold_1 = max(0.05, adj_prob_theshold_2 * 2)               # :163
    draws_1 = max(draws_after_tail / adj_prob_theshold_1, 1000)            # :166-167
    draws_2 = max(draws_after_tail / adj_prob_theshold_2, 1000)
    if approximate_posterior_analysis is None:                             # :170-176
        approximate_posterior_analysis = draws_2 > 20000

    # ---- format_input (R/utilities.R:924-959, :628-649)
    chk = data[data[do_check].astype(bool)]
    oth = data[~data[do_check].astype(bool)]
    sig_order = oth.sort_values(significance, kind="stable")[transcript].drop_duplicates()
    controls = set(sig_order.tail(how_many_negative_controls))
    my_df = pd.concat([chk, oth[oth[transcript].isin(controls)]], ignore_index=True)
    my_df = my_df[[transcript, sample, abundance] + covs + [do_check]].drop_duplicates()
    genes = list(dict.fromkeys(my_df[transcript]))                         # first-appearance order, checked first
    samples = list(dict.fromkeys(my_df[sample]))
    gidx = {g: i for i, g in enumerate(genes)}
    sidx = {s: i for i, s in enumerate(samples)}
    G, S = len(genes), len(samples)
    if len(my_df) != G * S:
        raise ValueError("The input data frame does not represent a rectangular structure. "
                         "Each transcript must be present in all samples.")
    counts = np.zeros((G, S), dtype=np.int32)
    counts[my_df[transcript].map(gidx).to_numpy(), my_df[sample].map(sidx).to_numpy()] = my_df[abundance].to_numpy()
    K = int(my_df.loc[my_df[do_check].astype(bool), transcript].nunique())

    X, xnames, x_samples = create_design_matrix(my_df, formula, sample)
    # The reference orders X by sorted sample name but S by first appearance; here the
    # design rows are put in S order so that both conventions agree.
    order = [x_samples.index(s) for s in samples]
    X = X[order]

    # ---- scaling (R/methods.R:222-238)
    if scaling_factor is None:
        mult, _ = get_scaled_counts_bulk(counts, samples)
    else:
        sf = data[[sample, scaling_factor]].drop_duplicates()
        mult = dict(zip(sf[sample], sf[scaling_factor]))
    multiplier = np.array([mult[s] for s in samples], dtype=np.float64)
    exposure_rate = -np.log(multiplier)

    from . import _lib
    # ---- enough memory for the full posterior? (R/methods.R:178-195 asks the host's RAM; the draws live on the device here)
    if not approximate_posterior_analysis:
        need = required_device_memory(G, X.shape[1], K, S, draws_2, cores, approximate_posterior_inference)
        free, _total = _lib.device_memory(device)
        if need > 0.9 * free:
            warnings.warn("You don't have enough memory to model the posterior distribution with MCMC draws. "
                          "Therefore the parameter approximate_posterior_analysis was set to TRUE")
            approximate_posterior_analysis = True
    multi = (devices is not None and len(devices) > 1 and not approximate_posterior_inference) or _pass is not None
    if _pass is not None:
        # one inference pass supplied by the caller -- ppcseq_amd.distributed.identify_outliers: the chains or the genes of a
        # pass over the ranks of a torch.distributed job -- in place of inference.do_inference; same arguments, same result type
        if approximate_posterior_inference:
            raise ValueError("passes over several ranks are NUTS passes (approximate_posterior_inference = False)")
    run_pass = _pass if _pass is not None else do_inference
    if multi and (save_generated_quantities or pass_fit):
        raise ValueError("devices=[...] deals the chains to several devices and pools their draws: not with save_generated_quantities / pass_fit")
    if devices is not None and len(devices) >= 1:
        device = devices[0]
    model = None if multi else _lib.Model(counts, X, exposure_rate, K, device=device)
    where = {} if _pass is not None else (dict(devices=list(devices)) if multi else dict(model=model))
    if _pass is None:
        where.update(approximate_posterior_inference=approximate_posterior_inference, pass_fit=pass_fit, launch=launch)
    try:
        # ---- pass 1: discovery (R/methods.R:268-286); always the full posterior analysis
        res1 = run_pass(counts, X, exposure_rate, K,
                            approximate_posterior_analysis=False, cores=cores,
                            adj_prob_theshold=adj_prob_theshold_1, how_many_posterior_draws=draws_1,
                            seed=seed, **where)
        if just_discovery:
            return res1.to_frame()
        # ---- cells to exclude (R/methods.R:292-300)
        flag = res1.deleterious_outliers if (do_check_only_on_detrimental and res1.deleterious_outliers is not None) else ~res1.ppc
        gg, ss = np.nonzero(flag)
        to_exclude = (gg * S + ss).astype(np.int32)
        # ---- pass 2: test (R/methods.R:320-342)
        if _pass is None:
            where.update(save_generated_quantities=save_generated_quantities)
        res2 = run_pass(counts, X, exposure_rate, K,
                            approximate_posterior_analysis=approximate_posterior_analysis, cores=cores,
                            adj_prob_theshold=adj_prob_theshold_2, how_many_posterior_draws=draws_2,
                            to_exclude=to_exclude, truncation_compensation=0.7352941,
                            seed=seed, **where)
    finally:
        if model is not None:
            model.close()

    # ---- merge_results / format_results (R/utilities.R:539-608)
    cov_by_sample = my_df[[sample] + covs].drop_duplicates().set_index(sample)
    rows = []
    for g in range(K):
        sw = pd.DataFrame({
            "S": np.arange(1, S + 1), "G": g + 1,
            abundance: counts[g], sample: samples,
            "slope_before_outlier_filtering": res1.slope[g],
            **{c: cov_by_sample.loc[samples, c].to_numpy() for c in covs},
            "exposure_rate": exposure_rate, "multiplier": multiplier,
            ".lower": res2.lower[g], ".upper": res2.upper[g],
            "slope_after_outlier_filtering": res2.slope[g],
            "posterior_predictive_check_succeded": res2.ppc[g],
        })
        if res2.deleterious_outliers is not None:
            sw["deleterious_outliers"] = res2.deleterious_outliers[g]
        row = {transcript: genes[g], "sample_wise_data": sw, "ppc_samples_failed": int((~res2.ppc[g]).sum())}
        if do_check_only_on_detrimental and res2.deleterious_outliers is not None:
            row["tot_deleterious_outliers"] = int(res2.deleterious_outliers[g].sum())
        rows.append(row)
    out = pd.DataFrame(rows)
    out.attrs.update(total_draws=res2.total_draws, transcript_column=transcript, abundance_column=abundance,
                     sample_column=sample, formula=formula, seed=seed,
                     diagnostics_discovery=res1.diagnostics, diagnostics_test=res2.diagnostics)
    if pass_fit:                                                           # R/methods.R:353-357: attrs "fit 1" / "fit 2"
        out.attrs["fit 1"], out.attrs["fit 2"] = res1.fit, res2.fit        # device-resident; the library keeps the model
    return out                                                             # alive until both fits are closed
