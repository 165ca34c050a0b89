#!/usr/bin/env python
"""Benchmark of the one hot path: NB hierarchical NUTS fit on MI355X (BASELINE.json metric
"effective samples/sec (whole node) for NB hierarchical fit").

A "step" is one complete fit -- Stan-default warm-up (150) + kept draws for every chain of every GPU --
on the synthetic 20,000 genes x 200 samples matrix of BASELINE config 3, inputs already resident in
HBM. `value` = sum over steps of the pooled bulk-ESS (min over the six hyper-parameters and lp__) divided
by the summed wall time (barrier + device synchronise on both sides, max over ranks). Chains are the
sharded unit (weak scaling: chains/GPU fixed); there is no collective on the data path. The timed fits run the
library's defaults (pipelined rounds, three chain groups on their own streams); the `roofline` sample comes from one
more fit on a single in-order stream, from HIP events attached to the merged launch's dispatches (hipExtLaunchKernel: the
kernel's own duration, as a kernel trace sees it).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=200)
    ap.add_argument("--chains-per-gpu", type=int, default=8,
                    help="chains per GPU; 8 = the chain count BASELINE cfg3 names, all on one GPU at N=1 (one launch of the "
                         "log-likelihood kernel then covers 8 chains: 2.4 -> 4.9 rounds of resident workgroups, a shorter tail)")
    ap.add_argument("--draws-per-chain", type=int, default=250)
    ap.add_argument("--nuts-warmup", type=int, default=150)
    ap.add_argument("--lanes", type=int, default=0, help="lanes per gene override (0 = automatic)")
    ap.add_argument("--workgroups", type=int, default=0, help="persistent workgroups of the log-likelihood launch (0 = automatic)")
    ap.add_argument("--mode", choices=["chains", "shards"], default="chains",
                    help="chains: BASELINE cfg3, chains partitioned over GPUs (default, weak scaling); shards: BASELINE cfg4 style, "
                         "genes partitioned over GPUs with an RCCL all-reduce of the partial sums every leapfrog (strong scaling)")
    ap.add_argument("--exchange", choices=["direct", "rccl"], default="direct",
                    help="shards mode: how the ranks' partial sums meet every leapfrog -- direct: peer-mapped buffers written by the "
                         "state machines inside the merged launch of a pipelined round (no collective call); rccl: an RCCL "
                         "all-reduce between the launches of the three-launch round")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-full-cfg2", action="store_true",
                    help="skip `cpu_baseline.measured`: ONE WHOLE fit of BASELINE cfg2 (5 000 x 50, 4 chains, 150 + 250) timed end to "
                         "end on the host's cores -- the optimised CPU comparator under the oracle's NUTS driver, same seed -- beside "
                         "the same fit on the GPU (about 20 s of CPU time; part of the default run since round 5)")
    ap.add_argument("--stream-groups", type=int, default=0,
                    help="chain groups on their own streams for the timed fits (0 = the library's default: 3 from eight chains on, 2 from four)")
    ap.add_argument("--single-stream-steps", type=int, default=1,
                    help="fits on ONE in-order stream (stream_groups = 1) after the timed ones: the source of the roofline's "
                         "per-launch timings, reported beside the headline (0 = skip: no roofline object)")
    ap.add_argument("--no-ppc", action="store_true", help="skip the posterior-predictive kernel's object")
    ap.add_argument("--as-named-steps", type=int, default=4,
                    help="fits of the configuration AS BASELINE cfg3 names it (1 chain per GPU), reported beside the headline "
                         "(0 = skip); outside the timed region of the headline")
    args = ap.parse_args()

    import torch
    from ppcseq_amd import _lib, distributed as D
    from ppcseq_amd.ess import ess_bulk
    from ppcseq_amd.synth import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    # one process per GPU; if a box has fewer GPUs than ranks (single-GPU rehearsal) ranks share devices
    n_dev = max(torch.cuda.device_count(), 1)
    dev_index = local_rank % n_dev
    backend = os.environ.get("PPCX_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm; "gloo" for rehearsals
    if dist_on:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        dist.init_process_group(backend, rank=rank, world_size=world)
    dev = f"cuda:{dev_index}" if backend == "nccl" else "cpu"   # where the collectives' tensors live
    torch.cuda.set_device(dev_index)

    G, S, C = args.genes, args.samples, 2
    seed_data = 20253
    # rank 0 draws the synthetic matrix; the other ranks receive it by RCCL broadcast over xGMI
    if rank == 0:
        d = synth(G, S, seed=seed_data, C=C)
        arrays = dict(counts=d["counts"], X=d["X"], exposure=d["exposure"], K=np.array([d["K"]], np.int64))
    else:
        arrays = None
    if dist_on:
        arrays = D.broadcast_arrays(arrays, device=dev)
    K = int(arrays["K"][0])
    comm, xchg = None, None
    if args.mode == "shards":
        # genes dealt to the ranks round-robin, as the reference deals them to its shards (R/utilities.R:125-136): every rank
        # gets its share of the K checked genes (which come first) and with them an equal share of the work
        if args.exchange == "rccl":             # the communicator's id travels from rank 0 through torch.distributed
            uid = [_lib.Comm.unique_id() if rank == 0 else None]
            if dist_on:
                import torch.distributed as dist
                dist.broadcast_object_list(uid, src=0)
            comm = _lib.Comm(world, rank, uid[0], device=dev_index)
        else:                                   # direct exchange: every rank maps every rank's receive buffer (IPC handles)
            xchg = _lib.Xchg(world, rank, args.chains_per_gpu, device=dev_index)
            if dist_on:
                import torch.distributed as dist
                handles = [None] * world
                dist.all_gather_object(handles, xchg.handle())
                xchg.connect(handles)
                dist.barrier()                  # nobody publishes before everybody has mapped everybody
        mine = np.arange(rank, G, world)
        model = _lib.Model(arrays["counts"][mine], arrays["X"], arrays["exposure"], 0, device=dev_index,
                           shard=(G, K, rank, None, world))
    else:
        model = _lib.Model(arrays["counts"], arrays["X"], arrays["exposure"], K, device=dev_index)
    if args.lanes or args.workgroups:
        model.set_launch(args.lanes, args.workgroups)
    Dm = model.D
    model_G = len(mine) if args.mode == "shards" else G
    hyper_cols = [0, 1, 2, Dm - 3, Dm - 2, Dm - 1]
    nch = args.chains_per_gpu
    n_iter = args.nuts_warmup + args.draws_per_chain

    def barrier():
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    if args.stream_groups > 0:
        model.set_rounds(stream_groups=args.stream_groups)
    elif args.mode == "shards":
        model.set_rounds(stream_groups=1)       # one in-order stream: the timed fits' launch timings are the roofline's sample

    def one_fit(step_seed):
        if comm is not None:                   # every rank runs the same chains on its genes
            return model.fit_nuts_comm(comm, chains=nch, iter=n_iter, warmup=args.nuts_warmup, seed=step_seed)
        if xchg is not None:
            return model.fit_nuts_xchg(xchg, chains=nch, iter=n_iter, warmup=args.nuts_warmup, seed=step_seed)
        return model.fit_nuts(chains=nch, iter=n_iter, warmup=args.nuts_warmup, seed=step_seed,
                              chain_id_offset=D.chain_id_offset(rank, nch))

    for w in range(args.warmup):
        f = one_fit(1000 + w)
        f.close()

    tot_time, tot_ess, tot_grad = 0.0, 0.0, 0
    kA_ms, kA_n, kA_chains = 0.0, 0, 0.0
    ppc_obj, rounds_last, single_kt = None, 0, None
    ess_detail, depth_mean, div_total = None, [], 0
    step_ms, stuck_fits = [], 0
    xchg_wait = (0.0, 0)
    for k in range(args.steps):
        barrier()
        t0 = time.perf_counter()
        fit = one_fit(1 + k)
        barrier()
        dt = time.perf_counter() - t0
        if dist_on:
            dt = D.max_over_ranks(dt, device=dev)
        # ---- outside the timed region: pooled ESS over every chain of the job
        hyp = fit.columns(hyper_cols)                      # [chains, n_keep, 6]
        dg = fit.diagnostics()
        tm = fit.timing()
        kt = fit.kernel_times()
        lp = dg["lp"]
        ge = np.array([float(tm.grad_evals)])
        if dist_on and args.mode == "chains":
            hyp = D.all_gather_chains(hyp, device=dev)
            lp = D.all_gather_chains(lp, device=dev)
            ge = D.all_gather_chains(ge, device=dev)
        per = [ess_bulk(hyp[:, :, j]) for j in range(6)] + [ess_bulk(lp)]
        # also the median over gene-level parameters (a sample of 128 intercepts and 128 sigma_raw; this rank's chains)
        gsel = np.unique(np.linspace(0, model_G - 1, 128).astype(int))
        gcols = [3 + int(g) for g in gsel] + [Dm - 3 - model_G + int(g) for g in gsel]
        gene_draws = fit.columns(gcols)
        ess_gene_median = float(np.median([ess_bulk(gene_draws[:, :, j]) for j in range(gene_draws.shape[2])]))
        tot_ess += float(np.nanmin(per))
        ess_detail = per
        tot_time += dt
        tot_grad += int(ge.sum())
        if args.mode == "shards":              # one stream there: the timed fits' own launch timings
            kA_ms += tm.gene_kernel_ms_mean * tm.gene_kernel_samples
            kA_n += tm.gene_kernel_samples
            kA_chains += tm.gene_kernel_chain_launches_mean * tm.gene_kernel_samples
        depth_mean.append(float(dg["treedepth"].mean()))
        step_ms.append(1e3 * dt)
        # a chain that left warm-up with a tiny step size runs every transition to the maximum tree depth and makes the fit
        # several times longer (2 of ~300 cfg3 fits): counted here, from this run's own fits
        stuck_fits += int((dg["treedepth"][:, args.nuts_warmup:].mean(axis=1) > 9.0).any())
        div_total += int(dg["divergent"][:, args.nuts_warmup:].sum())
        rounds_last = kt["launch_triples"]
        if xchg is not None:
            xw_us, xn = fit.xchg_timing()
            xchg_wait = (xchg_wait[0] + xw_us * xn, xchg_wait[1] + xn)
        # the posterior-predictive kernel on this fit's draws (outside the timed region; rank 0, last step)
        if rank == 0 and k == args.steps - 1 and not args.no_ppc and args.mode == "chains":
            ppc_obj = ppc_object(fit, K, S, C)
        fit.close()

    launch_used = model.get_launch()           # (lanes per gene, workgroups) of the timed fits' log-likelihood launches
    # BASELINE cfg3 as named: 8 chains, ONE per GPU (at N GPUs: N chains). A lone chain leaves the GPU mostly idle (a
    # leapfrog round costs the same for 1 chain as for 8), so this line scales strongly with nothing to gain from it; it is
    # reported for completeness, never as `value`.
    as_named = None
    if args.mode == "chains" and args.as_named_steps > 0:
        t_an, ess_an = 0.0, 0.0
        for k in range(args.as_named_steps):
            barrier()
            t0 = time.perf_counter()
            f1 = model.fit_nuts(chains=1, iter=n_iter, warmup=args.nuts_warmup, seed=501 + k, chain_id_offset=rank)
            barrier()
            dt = time.perf_counter() - t0
            if dist_on:
                dt = D.max_over_ranks(dt, device=dev)
            hyp1, lp1 = f1.columns(hyper_cols), f1.diagnostics()["lp"]
            f1.close()
            if dist_on:
                hyp1 = D.all_gather_chains(hyp1, device=dev)
                lp1 = D.all_gather_chains(lp1, device=dev)
            ess_an += float(np.nanmin([ess_bulk(hyp1[:, :, j]) for j in range(6)] + [ess_bulk(lp1)]))
            t_an += dt
        as_named = {"chains_per_gpu": 1, "chains_total": world, "value": round(ess_an / t_an, 3), "unit": "ESS/s",
                    "ms_per_step": round(1e3 * t_an / args.as_named_steps, 2), "steps": args.as_named_steps}

    # The same fit on ONE in-order stream: the merged launch's own start / stop events then time that launch alone, which is what
    # the roofline object needs (with chain groups on several streams another group's kernels share the chip with it).
    single = None
    if args.mode == "chains" and args.single_stream_steps > 0:
        model.set_rounds(stream_groups=1)
        try:
            t_g, ess_g = 0.0, 0.0
            for k in range(args.single_stream_steps):
                barrier()
                t0 = time.perf_counter()
                fg = one_fit(1 + k)
                barrier()
                dt = time.perf_counter() - t0
                if dist_on:
                    dt = D.max_over_ranks(dt, device=dev)
                hypg, lpg = fg.columns(hyper_cols), fg.diagnostics()["lp"]
                tmg, single_kt = fg.timing(), fg.kernel_times()
                fg.close()
                kA_ms += tmg.gene_kernel_ms_mean * tmg.gene_kernel_samples
                kA_n += tmg.gene_kernel_samples
                kA_chains += tmg.gene_kernel_chain_launches_mean * tmg.gene_kernel_samples
                if dist_on:
                    hypg = D.all_gather_chains(hypg, device=dev)
                    lpg = D.all_gather_chains(lpg, device=dev)
                ess_g += float(np.nanmin([ess_bulk(hypg[:, :, j]) for j in range(6)] + [ess_bulk(lpg)]))
                t_g += dt
            single = {"stream_groups": 1, "chains_total": nch * world, "value": round(ess_g / t_g, 3), "unit": "ESS/s",
                      "ms_per_step": round(1e3 * t_g / args.single_stream_steps, 2), "steps": args.single_stream_steps,
                      "kernel_ms": {k: round(v, 5) for k, v in single_kt.items()}}
        finally:
            model.set_rounds(stream_groups=args.stream_groups)

    if rank == 0:
        E = 0
        b_grad = 4.0 * G * S + 16.0 * (C + 1) * G + 8.0 * S * (C + 1) + 4.0 * E
        roof = None
        if kA_n > 0:
            ms = kA_ms / kA_n
            chains_per_launch = kA_chains / kA_n
            if args.mode == "shards":
                b_grad = b_grad / world              # each rank streams its share of the genes
            achieved = b_grad * chains_per_launch / (ms * 1e-3) / 1e9
            piped = model.get_rounds(nch)[0] and comm is None
            roof = {"bound": "hbm", "kernel": "ppcx_ls_kernel" if piped else "ppcx_loglik_kernel", "achieved": round(achieved, 2), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": pmc_traffic(chains_per_launch), "fp64_issue": issue_profile(),
                    "algorithmic_bytes_per_launch": b_grad * chains_per_launch, "avg_launch_ms": round(ms, 5),
                    "timed_launches": int(kA_n),
                    "measured_by_this_run": ["achieved", "frac", "avg_launch_ms", "timed_launches"],
                    "from_committed_profiles": {"traffic": "profiles/rNN_pmc.json (rocprofv3 --pmc passes of the same command: counters need their own runs)",
                                                "fp64_issue": "profiles/rNN_loglik_issue.json (SQ counters; `stale` says whether the kernel sources have changed since)"},
                    "sampled_in": "a fit on one in-order stream (stream_groups = 1) after the timed fits" if args.mode == "chains" else "the timed fits",
                    "timed_by": "HIP events attached to the dispatch of sampled launches with every chain active (hipExtLaunchKernel start / stop events)",
                    "note": "achieved = algorithmic bytes (SURVEY 8d: count matrix + coordinates, per chain gradient) x chains per "
                            "launch / launch time: an effective-throughput figure. ppcx_ls_kernel is the merged launch of a "
                            "pipelined round: the log-likelihood workgroups of every chain beside the chains' state machines. "
                            "The chains of a launch share the count matrix through L2 / Infinity Cache (traffic = FETCH_SIZE + "
                            "WRITE_SIZE of the PMC passes is far below the algorithmic bytes), and the kernel is bound by fp64 "
                            "vector issue, not by HBM (profiles/ README: instructions per cell, VALU busy share)."}
        cpu = None
        if not args.no_cpu_baseline and world == 1:        # reported at N = 1 only (rank 0's host cores)
            cpu = cpu_baseline(arrays, K, tot_ess, tot_grad, args)
        out = {
            "metric": "effective samples/sec (whole node) for NB hierarchical fit",
            "value": round(tot_ess / tot_time, 3), "unit": "ESS/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * tot_time / max(args.steps, 1), 2),
            "higher_is_better": True, "scaling": "weak" if args.mode == "chains" else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE cfg3: synthetic {G} genes x {S} samples (seed {seed_data}), C=2, K={K}; "
                                    f"NUTS (Stan defaults) warm-up {args.nuts_warmup} + {args.draws_per_chain} kept draws/chain, "
                                    f"{nch} chains per GPU (cfg3 names 8 chains; chains are the sharded unit, so N GPUs run {nch}N chains)") if args.mode == "chains" else
                                   (f"BASELINE cfg4 style: synthetic {G} genes x {S} samples (seed {seed_data}), C=2, K={K}, genes "
                                    f"sharded over {world} GPU(s) with an RCCL all-reduce of the partial sums per leapfrog; NUTS "
                                    f"(Stan defaults) warm-up {args.nuts_warmup} + {args.draws_per_chain} kept draws/chain, {nch} chains"),
                       "mode": args.mode, "chains_total": nch * world if args.mode == "chains" else nch, "lanes_per_gene": launch_used[0], "loglik_workgroups": launch_used[1],
                       "ess_estimator": "rank-normalised split-chain bulk-ESS, min over 6 hyper-parameters and lp__",
                       "ess_last_step": [round(float(x), 1) for x in ess_detail],
                       "ess_median_intercept_sigma_raw_rank0_chains": round(ess_gene_median, 1),
                       "grad_evals": tot_grad, "mean_treedepth": round(float(np.mean(depth_mean)), 2),
                       "divergent_after_warmup": div_total,
                       "ms_per_step_min_max": [round(min(step_ms), 1), round(max(step_ms), 1)] if step_ms else None,
                       "timed_fits_with_a_chain_at_max_treedepth_rank0": stuck_fits,
                       "round_structure": ("pipelined: merged log-likelihood / state-machine launch + gene kernel" if (model.get_rounds(nch)[0] and comm is None)
                                           else "three launches: log-likelihood, close, step + update"),
                       "stream_groups": args.stream_groups if args.stream_groups > 0 else ("library default (3 from eight chains on, 2 from four)" if args.mode == "chains" else 1),
                       "rounds_last_step_all_groups": int(rounds_last),
                       "us_per_grad_eval_per_chain": round(1e6 * tot_time * world * nch / max(tot_grad, 1), 2),
                       **({"exchange": args.exchange,
                           "exchange_us_per_round": (round(xchg_wait[0] / xchg_wait[1], 3) if xchg_wait[1] else (0.0 if world == 1 else None)),
                           "exchange_note": "direct: mean time a chain's state machine waited for its peers' sums per round, measured in the kernel "
                                            "(100 MHz wall clock), rank 0; it runs beside the log-likelihood workgroups of the same launch, so it is "
                                            "not added to the round. One rank: no exchange takes place"} if args.mode == "shards" else {})},
            "roofline": roof, "ppc": ppc_obj, "cpu_baseline": cpu, "as_named_cfg3_one_chain_per_gpu": as_named, "single_stream": single,
            "concordance": None if (args.no_cpu_baseline or world > 1) else outlier_concordance(),
        }
        if cpu is not None and not args.no_cpu_full_cfg2 and world == 1:
            # the one CPU / GPU pair that is MEASURED end to end (whole fits of cfg2), beside the extrapolated cfg3 figure
            full = cpu_full_cfg2(dev_index)
            cpu["measured"] = {"config": "cfg2", "workload": full["workload"], "cpu_s": full["cpu"]["seconds"], "gpu_s": full["gpu"]["seconds"],
                               "cpu_ess_per_s": full["cpu"]["ess_per_s"], "gpu_ess_per_s": full["gpu"]["ess_per_s"],
                               "gpu_over_cpu_ess_per_s": full["gpu_over_cpu_ess_per_s"], "cores": full["cpu"]["cores"],
                               "threads": full["cpu"]["threads"], "cpu_grad_evals": full["cpu"]["grad_evals"],
                               "gpu_grad_evals": full["gpu"]["grad_evals"], "same_first_trees": full["same_first_trees"],
                               "kind": full["cpu"]["kind"], "extrapolated": False}
        print(json.dumps(out))
    model.close()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(chains_per_launch):
    """HBM-side bytes per launch of the log-likelihood kernel from the latest committed PMC passes (profiles/rNN_pmc.json,
    written by scripts/profile_round.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs). PMC counters
    cannot be read inside this process, so the figure belongs to the profiled run of this same command; it is reported
    only when that run had the same number of chains per launch, else null."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            p = json.load(fh)
        if int(p.get("chains_per_launch", 0)) != int(round(chains_per_launch)):
            return None
        return round((p["fetch_size_kib_per_launch"] + p["write_size_kib_per_launch"]) * 1024.0, 1)
    except Exception:
        return None


def issue_profile():
    """What actually bounds the log-likelihood kernel -- fp64 vector issue -- from the latest committed counter profile
    (profiles/rNN_loglik_issue.json: SQ counters, clock and arithmetic-ceiling microbenchmarks of this kernel at the
    headline workload). Reported inside `roofline` beside the HBM figures; null when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_loglik_issue.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as fh:
            p = json.load(fh)
        out = {k: p[k] for k in ("vector_insts_per_cell_iteration", "vector_pipes_busy_frac_at_sustained_clock",
                                 "sustained_clock_ghz_under_fp64_fma", "frac_of_cell_arithmetic_ceiling") if k in p} | {"profile": os.path.basename(files[-1])}
        # the counters belong to the kernel sources they were taken on: say so when those have changed since
        out["kernel_sources_sha16"] = p.get("kernel_sources_sha16")
        out["stale"] = p.get("kernel_sources_sha16") != kernel_sources_sha16()
        return out
    except Exception:
        return None


def kernel_sources_sha16():
    """Hash of the HIP sources the log-likelihood kernel is built from (recorded in profiles/rNN_loglik_issue.json)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("ppcx_math.h", "ppcx_disp.h", "ppcx_model.h", "ppcx_nuts.h", "ppcx_gene.h", "ppcx_kernels.h", "ppcx_kernels.hip"):
        with open(os.path.join(ROOT, "ppcseq_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def ppc_object(fit, K, S, C):
    """The posterior-predictive kernel (inst/stan/negBinomial_MPI.stan:259-266 + the summary of R/utilities.R:685-703) on the
    kept draws of a timed fit: every checked cell (K x S) draws one negative-binomial count per kept posterior draw, keeps
    them in LDS and reduces them to mean / sd / two type-7 quantiles. Algorithmic bytes per posterior draw:
    4 S K (the counts_rng the reference materialises) + 8 (S + K (C + 1)) (exposure and the K-slice of the parameters)."""
    n_draws = fit.chains * fit.n_keep
    fit.ppc(1.0, 0.025, 0.975, seed=1)                               # warm-up launch (code objects, clocks)
    ms = []
    for _ in range(3):
        fit.ppc(1.0, 0.025, 0.975, seed=1)
        ms.append(fit.ppc_timing()[0])
    t = min(ms) * 1e-3
    nb = fit.ppc_timing()[1]
    b_draw = 4.0 * S * K + 8.0 * (S + K * (C + 1))
    achieved = b_draw * n_draws / t / 1e9
    return {"kernel": "ppcx_ppc_table_kernel + " + ("ppcx_ppc_wave_kernel" if n_draws <= 4096 else "ppcx_ppc_kernel"),
            "workload": f"{K} checked genes x {S} samples x {n_draws} kept draws of the timed fit",
            "nb_draws": int(nb), "kernel_ms": round(1e3 * t, 3), "nb_draws_per_s": round(nb / t, 1),
            "algorithmic_bytes_per_posterior_draw": b_draw, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": round(achieved / 8000.0, 6), "bound": "alu",
            "note": "not memory bound: a negative-binomial draw is 5-7 Philox blocks (quarter-rate 32 x 32 multiplies), a "
                    "Marsaglia-Tsang gamma and a Knuth / PTRS Poisson -- ~1500 vector-instruction slots per wavefront-draw against 4 "
                    "bytes (profiles/ README: SQ counters of this kernel). One wavefront per cell, parameters from a transposed "
                    "table, one loop of sampler attempts per wavefront; the draws never leave LDS unless save_generated_quantities "
                    "asks for them, so the 4 S K bytes of the reference's counts_rng are not written at all"}


def outlier_concordance():
    """Second half of the BASELINE metric: outlier-call concordance. The reference's test configuration (bundled
    `counts`: SLC16A12 / CYP1A1 / ART3 + 50 negative controls, ~ Label, pfp = 1; tests/testthat/test-ppcSeq.R:11-24),
    discovery pass (3 chains, 150 + 334 iterations, 5 %/95 % interval): flags from the GPU path vs flags from the CPU
    oracle (restatement of the Stan path; no rstan on this box) at the same seed."""
    fx = os.path.join(ROOT, "tests", "golden", "counts_bundled.npz")
    if not os.path.exists(fx):
        return None
    from oracle.oracle import Oracle
    from ppcseq_amd.inference import do_inference, _post_process
    from ppcseq_amd.methods import get_scaled_counts_bulk
    z = np.load(fx)
    genes = [str(g) for g in z["genes"]]
    checked = [genes.index(g) for g in ("SLC16A12", "CYP1A1", "ART3")]
    others = [i for i in range(len(genes)) if i not in checked]
    ctrl = set(sorted(others, key=lambda i: z["PValue"][i])[-50:])
    sel = checked + [i for i in others if i in ctrl]
    counts = z["value"][sel].astype(np.int32)
    lab = z["Label"].astype(str)
    X = np.stack([np.ones(len(lab)), (lab == sorted(set(lab))[1]).astype(float)], axis=1)
    mult, _ = get_scaled_counts_bulk(counts, list(range(counts.shape[1])))
    expo = -np.log(np.array([mult[s] for s in range(counts.shape[1])]))
    seed, p = 42, 0.05
    g = do_inference(counts, X, expo, 3, cores=1, adj_prob_theshold=p, how_many_posterior_draws=1000, seed=seed)
    O = Oracle()
    mo = O.model(counts, X, expo, 3, n_threads=4)
    r = O.nuts_model(mo, O.cfg(chains=3, iter=g.iter, warmup=150, seed=seed))
    dr = r.draws.reshape(-1, r.draws.shape[-1])
    gq = O.generated_quantities(mo, dr, 1.0, seed=seed)          # [draws, K, S]
    ci = O.summarise(gq, p, 1 - p)
    # Monte-Carlo standard error of the upper interval end per cell (bootstrap over the draws of the CPU path); the two
    # paths are independent runs, so their difference has sqrt(2) times that error
    rng = np.random.default_rng(0)
    nd = gq.shape[0]
    boots = np.stack([np.quantile(gq[rng.integers(0, nd, nd)], 1 - p, axis=0) for _ in range(100)])
    se_upper = np.sqrt(2.0) * boots.std(axis=0) + 0.5              # + half a count: the ends are interpolated integers
    off = 3 + counts.shape[0]
    o = _post_process(counts[:3], ci, dr[:, off:off + 3].mean(0), X)
    same_ppc = float(np.mean(g.ppc == o.ppc))
    same_del = float(np.mean(g.deleterious_outliers == o.deleterious_outliers))
    return {"config": "bundled counts, 3 checked genes + 50 controls, ~Label, discovery pass, 3 chains x (150+334), seed 42",
            "cells": int(g.ppc.size), "ppc_identical": same_ppc, "deleterious_outliers_identical": same_del,
            "gpu_tot_deleterious": [int(v) for v in g.deleterious_outliers.sum(1)],
            "cpu_tot_deleterious": [int(v) for v in o.deleterious_outliers.sum(1)],
            "max_upper_ci_rel_diff": float(np.max(np.abs(g.upper - o.upper) / (1 + o.upper))),
            "median_upper_ci_rel_diff": float(np.median(np.abs(g.upper - o.upper) / (1 + o.upper))),
            "max_upper_ci_diff_in_mc_standard_errors": float(np.max(np.abs(g.upper - o.upper) / se_upper))}


def cpu_full_cfg2(dev_index, seed=1):
    """One WHOLE fit of BASELINE cfg2 (synthetic 5 000 genes x 50 samples, 4 chains, warm-up 150 + 250 kept draws) timed end to
    end on the host's cores beside the same fit on the GPU -- same data, same sampler seed and chain ids (hence the same Philox
    streams), same ESS estimator -- so that one CPU / GPU pair of this repository is measured, not extrapolated. The CPU side is
    the oracle's NUTS driver (Stan-default sampler, oracle/ppc_oracle.c run_chain) on the optimised comparator's gradient
    (oracle/cpu_fast.cpp), one host thread per chain as rstan runs chains on `cores` workers (R/utilities.R:1500-1501) and
    cores / chains OpenMP threads over genes inside a gradient (map_rect shards, R/utilities.R:1383-1386,1479).
    "CPU restatement of this repository, not rstan"."""
    from oracle.oracle import CpuFast, Oracle
    from ppcseq_amd import _lib
    from ppcseq_amd.ess import ess_bulk
    from ppcseq_amd.synth import synth
    chains, warm, keep = 4, 150, 250
    d = synth(5000, 50, seed=20252)
    try:
        cores = min(len(os.sched_getaffinity(0)), 16)
    except AttributeError:
        cores = min(os.cpu_count() or 1, 16)

    def ess_of(draws, lp):
        D = draws.shape[-1]
        hy = draws[:, :, [0, 1, 2, D - 3, D - 2, D - 1]]
        return float(np.nanmin([ess_bulk(hy[:, :, j]) for j in range(6)] + [ess_bulk(lp)]))

    m = _lib.Model(d["counts"], d["X"], d["exposure"], d["K"], device=dev_index)
    f = m.fit_nuts(chains=chains, iter=warm + keep, warmup=warm, seed=seed + 1000)     # untimed: first-launch costs
    f.close()
    t0 = time.perf_counter()
    f = m.fit_nuts(chains=chains, iter=warm + keep, warmup=warm, seed=seed)
    t_gpu = time.perf_counter() - t0
    dg = f.diagnostics()
    ess_gpu, grads_gpu = ess_of(f.draws(), dg["lp"]), int(dg["n_leapfrog"].sum())
    f.close(); m.close()
    O, F = Oracle(), CpuFast()
    cfg = O.cfg(chains=chains, iter=warm + keep, warmup=warm, seed=seed)
    tpc = max(1, cores // chains)
    t0 = time.perf_counter()
    r = F.nuts(O, d["counts"], d["X"], d["exposure"], d["K"], cfg, threads_per_chain=tpc)
    t_cpu = time.perf_counter() - t0
    ess_cpu, grads_cpu = ess_of(r.draws, r.lp), int(r.n_leapfrog.sum())
    return {"workload": "BASELINE cfg2: synthetic 5000 genes x 50 samples (seed 20252), 4 chains, warm-up 150 + 250 kept draws, sampler seed %d" % seed,
            "gpu": {"seconds": round(t_gpu, 3), "ess_min": round(ess_gpu, 1), "ess_per_s": round(ess_gpu / t_gpu, 2), "grad_evals": grads_gpu},
            "cpu": {"seconds": round(t_cpu, 2), "ess_min": round(ess_cpu, 1), "ess_per_s": round(ess_cpu / t_cpu, 3), "grad_evals": grads_cpu,
                    "cores": cores, "threads": chains * tpc, "chains_in_parallel": chains, "threads_per_chain": tpc,
                    "kind": "port-optimised (oracle/cpu_fast.cpp under the oracle's NUTS driver); not rstan"},
            "gpu_over_cpu_ess_per_s": round((ess_gpu / t_gpu) / (ess_cpu / t_cpu), 1),
            "same_first_trees": bool(np.array_equal(dg["n_leapfrog"][:, :6], r.n_leapfrog[:, :6])),
            "extrapolated": False}


def cpu_baseline(arrays, K, gpu_ess, gpu_grad, args):
    """Two CPU comparators timed on this host on a bounded sample of the same workload -- gradient evaluations of the
    20 000 x 200 model at a fixed point for about --cpu-seconds in total, OpenMP threads over genes (mirrors map_rect /
    STAN_NUM_THREADS, R/utilities.R:1383-1386,1479), built -O3 -march=native:
      "port"            the oracle, a literal restatement of the Stan program (libm lgamma, series digamma, log1p_exp per cell);
      "port-optimised"  the product's own formulation compiled for the host (oracle/cpu_fast.cpp: sufficient statistics, one
                        table logarithm and one reciprocal per cell, Stirling tails) -- the faster one, quoted as `value`.
    Neither is rstan (no R / Stan on this box). A whole fit is ~3e5 gradient evaluations, so `value` is an EXTRAPOLATION:
    gradient evaluations per second x the effective samples per gradient evaluation of the GPU run (same algorithm, seeds
    and estimator)."""
    from oracle.oracle import CpuFast, Oracle
    try:
        O = Oracle(native=True)
        build = "-O3 -march=native"
    except Exception:
        O = Oracle()
        build = "-O3 -march=x86-64-v3"
    # the GPU box gives a one-GPU job a share of ~16 host cores whatever os.cpu_count() says
    try:
        cores = min(len(os.sched_getaffinity(0)), 16)
    except AttributeError:
        cores = min(os.cpu_count() or 1, 16)
    counts, X, expo = arrays["counts"], arrays["X"], arrays["exposure"]
    G, S = counts.shape
    cells = float(G) * S
    ess_per_grad = gpu_ess / max(gpu_grad, 1)
    budget = max(float(args.cpu_seconds), 2.0) / 2.0

    def rate(fn):
        fn()                                                     # page the matrix in
        t0 = time.perf_counter()
        n = 0
        while True:
            fn()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget:
                return n / dt, n, dt

    m = O.model(counts, X, expo, K, n_threads=cores)
    u = np.zeros(O.dim(m.G, m.C, m.K))
    u[3:3 + G] = 5.0
    r_port, n_port, t_port = rate(lambda: O.log_prob_grad(m, u))
    lp_port, g_port = O.log_prob_grad(m, u)
    fast = None
    try:
        F = CpuFast()
        mf = F.model(counts, X, expo, K)
        r_fast, n_fast, t_fast = rate(lambda: F.log_prob_grad(mf, u, threads=cores))
        lp_fast, g_fast = F.log_prob_grad(mf, u, threads=cores)
        F.free(mf)
        fast = {"grad_evals_per_s": round(r_fast, 3), "ns_per_cell_per_thread": round(1e9 * cores / (r_fast * cells), 2),
                "grad_evals_sampled": n_fast, "seconds_sampled": round(t_fast, 1), "value": round(r_fast * ess_per_grad, 5),
                "agrees_with_port": {"lp_rel": float(abs(lp_fast - lp_port) / abs(lp_port)),
                                     "grad_rel_max": float(np.max(np.abs(g_fast - g_port) / (1 + np.abs(g_port))))}}
    except Exception as e:                                       # no g++ / OpenMP on the box: the literal port alone
        fast = {"error": repr(e)}
    port = {"grad_evals_per_s": round(r_port, 3), "ns_per_cell_per_thread": round(1e9 * cores / (r_port * cells), 1),
            "grad_evals_sampled": n_port, "seconds_sampled": round(t_port, 1), "value": round(r_port * ess_per_grad, 5)}
    best_kind = "port-optimised" if "value" in fast else "port"
    best = fast if "value" in fast else port
    return {"value": best["value"], "unit": "ESS/s", "cores": cores, "kind": best_kind, "build": build, "extrapolated": True,
            "grad_evals_per_s": best["grad_evals_per_s"], "ns_per_cell_per_thread": best["ns_per_cell_per_thread"],
            "port": port, "port_optimised": fast,
            "sample": f"gradient evaluations of the same {G}x{S} model at a fixed point, {cores} OpenMP threads over genes: "
                      f"{port['grad_evals_sampled']} by the literal port in {port['seconds_sampled']} s"
                      + (f", {fast['grad_evals_sampled']} by the optimised comparator in {fast['seconds_sampled']} s" if "value" in fast else "")
                      + "; value = gradient evaluations per second x ESS per gradient evaluation of the GPU run (same algorithm, seeds "
                        "and estimator). CPU restatements of this repository (Stan-equivalent), not rstan."}


if __name__ == "__main__":
    main()
