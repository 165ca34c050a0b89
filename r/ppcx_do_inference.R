# r/ppcx_do_inference.R -- the R-side shim of the drop-in boundary. Inside do_inference() it replaces
#   fit = switch(approximate_posterior_inference %>% `!` %>% as.integer %>% sum(1),
#                vb_iterative(...), sampling(...))                       (R/utilities.R:1482-1513)
# together with the read-outs of the fit that follow it: fit_to_counts_rng / fit_to_counts_rng_approximated
# (R/utilities.R:685-703, :733-784) and summary_to_tibble(fit, "alpha_sub_1") (:1531, :1250-1263).
# One .C() call = one inference pass. `dims[1]` is the ABI version this shim was written for (include/ppcx.h
# PPCX_VERSION): the library refuses any other with status -1 before reading anything else.
ppcx_do_inference <- function(counts_GS, X, exposure_rate, K, to_exclude_cells,
                              approximate_posterior_inference,      # TRUE = ADVI, the reference's default (R/methods.R:85)
                              chains, iter, warmup = 150L, seed,
                              adj_prob_theshold, truncation_compensation,
                              how_many_posterior_draws, approximate_posterior_analysis,
                              save_generated_quantities = FALSE,
                              lambda_mu_mu = 5.612671, device = 0L,
                              devices = integer(0)) {      # several GPUs: the chains of a NUTS pass are dealt to them, as
                                                           # sampling(cores = ...) deals them to its workers (R/utilities.R:1500-1501)
  G <- nrow(counts_GS); S <- ncol(counts_GS); C <- ncol(X)
  draws_practical <- if (approximate_posterior_analysis) 1000L else as.integer(how_many_posterior_draws)  # R/utilities.R:1372
  n_gen <- if (approximate_posterior_analysis) as.integer(how_many_posterior_draws) else 0L
  n_draws_out <- if (n_gen > 0L) n_gen else if (approximate_posterior_inference) draws_practical else chains * (iter - warmup)
  stopifnot(length(devices) <= 16L)
  dims <- as.integer(c(400L,                               # PPCX_VERSION this shim was written for (checked by the library)
                       device, G, S, C, K, length(to_exclude_cells), chains, iter, warmup,
                       n_gen, as.integer(approximate_posterior_analysis),
                       as.integer(approximate_posterior_inference), as.integer(save_generated_quantities),
                       draws_practical,                   # vb output_samples (R/utilities.R:1490)
                       50000L,                            # vb iter (R/utilities.R:1491)
                       length(devices), devices, rep(0L, 16L - length(devices))))   # n_devices, devices[16]
  reals <- c(lambda_mu_mu, truncation_compensation, adj_prob_theshold, 1 - adj_prob_theshold, seed,
             0.005)                                       # vb tol_rel_obj, hard-coded by the reference (R/utilities.R:1492)
  out <- .C("ppcx_do_inference_C",
            dims    = dims,
            counts  = as.integer(t(counts_GS)),          # gene-major, sample index fastest
            X       = as.double(X),                      # column-major = R native
            expo    = as.double(exposure_rate),
            excl    = as.integer(to_exclude_cells),      # 0-based g*S+s
            reals   = as.double(reals),
            ci      = double(K * S * 4L),                # [g][s][mean, sd, .lower, .upper]
            slope   = double(K),
            counts_rng = integer(if (save_generated_quantities) n_draws_out * K * S else 1L),   # [draw][g][s]
            status  = integer(1L),
            errbuf  = strrep(" ", 255L),                 # .C passes character vectors as char**
            errlen  = 256L)
  if (out$status != 0L) stop(sprintf("ppcx error %d: %s", out$status, trimws(out$errbuf)))  # NUTS errors propagate, as in rstan
  ci <- aperm(array(out$ci, c(4L, S, K)), c(3L, 2L, 1L)) # K x S x 4
  list(ci = ci, slope = out$slope,
       counts_rng = if (save_generated_quantities) aperm(array(out$counts_rng, c(S, K, n_draws_out)), c(3L, 2L, 1L)))
}

# What do_inference() does with the result: the tibble the rest of the package expects (R/utilities.R:1516-1544).
# `my_df`, `.do_check`, `.abundance`, `X`, `how_many_to_check`, `S` are the variables of do_inference's frame.
ppcx_fit_to_tibble <- function(res, my_df, .do_check, .abundance, X, K, S) {
  fit_tbl <- tidyr::expand_grid(G = seq_len(K), S = seq_len(S)) %>%
    dplyr::mutate(.variable = "counts_rng",
                  mean = as.vector(t(res$ci[, , 1])), sd = as.vector(t(res$ci[, , 2])),
                  .lower = as.vector(t(res$ci[, , 3])), .upper = as.vector(t(res$ci[, , 4])))
  fit_tbl %>%
    check_if_within_posterior(my_df, .do_check, .abundance) %>%                 # R/utilities.R:651-663, unchanged
    dplyr::left_join(tibble::tibble(G = seq_len(K), slope = res$slope), by = "G") %>%   # replaces summary_to_tibble(fit, "alpha_sub_1")
    add_deleterious_if_covariate_exists(X)                                      # R/utilities.R:493-513, unchanged
}
