"""Multi-GPU layer: one process per GPU, chains partitioned across ranks.

Mirrors the reference's chain parallelism (`chains=`/`cores=` of rstan::sampling, R/utilities.R:1500-1501,
which forks one R worker per chain): chains are independent units, so there is NO per-leapfrog
collective. torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests)
is used for plumbing only: rank 0 broadcasts the model inputs, and the kept draws of the checked genes /
diagnostics are all-gathered for the pooled summaries (`do_inference`: credible intervals and flags from the merged
chains of all ranks, as rstan::summary gives them, R/utilities.R:685-703).
"""
from __future__ import annotations

import numpy as np


def chain_id_offset(rank: int, chains_per_rank: int) -> int:
    """Global id of this rank's first chain: the Philox key of chain c is (seed, c), so every chain in
    the job has its own stream wherever it runs."""
    return rank * chains_per_rank


def _dist():
    import torch.distributed as dist
    return dist


def is_initialized() -> bool:
    try:
        d = _dist()
        return d.is_available() and d.is_initialized()
    except Exception:
        return False


def broadcast_arrays(arrays: dict | None, device="cpu", src=0):
    """Broadcast a dict of numpy arrays from `src` to all ranks (model inputs: counts, X, exposure)."""
    import torch
    dist = _dist()
    meta = [None]
    if dist.get_rank() == src:
        meta[0] = {k: (v.shape, str(v.dtype)) for k, v in arrays.items()}
    dist.broadcast_object_list(meta, src=src)
    out = {}
    for k, (shape, dtype) in meta[0].items():
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(arrays[k])).to(device)
        else:
            t = torch.empty(shape, dtype=getattr(torch, dtype.replace("float64", "float64")), device=device)
        dist.broadcast(t, src=src)
        out[k] = t.cpu().numpy()
    return out


def all_gather_chains(x: np.ndarray, device="cpu") -> np.ndarray:
    """Concatenate per-rank arrays [chains_local, ...] along axis 0 in rank order."""
    import torch
    dist = _dist()
    t = torch.from_numpy(np.ascontiguousarray(x)).to(device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return np.concatenate([p.cpu().numpy() for p in parts], axis=0)


def max_over_ranks(value: float, device="cpu") -> float:
    import torch
    dist = _dist()
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def raise_if_any_rank_failed(err: BaseException | None, device="cpu", what="a rank"):
    """Collective error check: every rank passes its own exception (or None). If any rank failed, EVERY rank raises -- the
    failing ones their own exception, the others a RuntimeError naming the lowest failing rank -- instead of the healthy
    ranks blocking in the next collective for a peer that has already left."""
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    worst = max_over_ranks(float(world - rank) if err is not None else 0.0, device=device)     # > 0: the lowest failing rank
    if worst > 0.0:
        if err is not None:
            raise err
        raise RuntimeError(f"{what}: rank {world - int(worst)} failed; this rank leaves with it")


def any_rank_failed(err: BaseException | None, device="cpu") -> bool:
    """Collective: whether any rank passed an exception (nobody raises -- the caller has a way on for all ranks together)."""
    return max_over_ranks(1.0 if err is not None else 0.0, device=device) > 0.0


def do_inference(counts, X, exposure_rate, how_many_to_check, *, device=0, coll_device="cpu", chains=None, cores=None,
                 approximate_posterior_analysis=False, lambda_mu_mu=5.612671, adj_prob_theshold=0.05,
                 how_many_posterior_draws=1000, to_exclude=None, truncation_compensation=1.0, seed=1, launch=None):
    """One inference pass of ppcseq (ppcseq_amd.inference.do_inference) with the chains partitioned over the ranks of the
    initialised torch.distributed job, one process per GPU. Every rank holds the inputs, fits its chains with the global
    chain ids (the Philox stream of chain c does not depend on the rank that runs it), the draws of the checked genes are
    all-gathered (a few MB) and rank 0 computes the credible intervals and flags from the POOLED draws; the result is
    broadcast, so every rank returns the same InferenceResult as one fit of all the chains would give.
    `launch` = (lanes_per_gene, workgroups) pins the log-likelihood launch (0 = automatic) (bit-identical results across rank counts need
    the same geometry: the automatic choice depends on the chains per launch)."""
    import math
    from . import _lib
    from .inference import _to_cell_ids, checked_columns, find_optimal_number_of_chains, pooled_summary
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    counts = np.asarray(counts)
    G, S = counts.shape
    K = int(how_many_to_check)
    X = np.asarray(X, dtype=np.float64).reshape(S, -1)
    practical = 1000 if approximate_posterior_analysis else how_many_posterior_draws
    if chains is None:
        chains = max(3, min(int(cores) if cores else world, find_optimal_number_of_chains(practical)))
    n_iter = int(math.ceil(practical / chains)) + 150
    per = int(math.ceil(chains / world))
    n_local = max(0, min(per, chains - rank * per))
    cols = checked_columns(G, X.shape[1], K)
    n_keep = n_iter - 150
    local = np.zeros((per, n_keep, cols.size))
    err = None
    if n_local > 0:
        try:
            m = _lib.Model(counts, X, exposure_rate, K, lambda_mu_mu=lambda_mu_mu, excl=_to_cell_ids(to_exclude, S), device=device)
            try:
                if launch is not None:
                    m.set_launch(*launch)
                f = m.fit_nuts(chains=n_local, iter=n_iter, warmup=150, seed=seed, chain_id_offset=rank * per)
                try:
                    local[:n_local] = f.columns(cols)
                finally:
                    f.close()
            finally:
                m.close()
        except Exception as e:                     # out of memory, no finite initial point, a limit of this build ...
            err = e
    raise_if_any_rank_failed(err, device=coll_device, what="chains over ranks")      # before anybody waits in the gather
    pooled = all_gather_chains(local, device=coll_device)[:chains]        # ranks past the last chain contributed padding
    res, err = [None], None
    if rank == 0:
        try:
            r = pooled_summary(counts, X, exposure_rate, K, pooled, lambda_mu_mu=lambda_mu_mu,
                               approximate_posterior_analysis=approximate_posterior_analysis, adj_prob_theshold=adj_prob_theshold,
                               how_many_posterior_draws=how_many_posterior_draws, truncation_compensation=truncation_compensation,
                               seed=seed, device=device)
            r.chains, r.iter = chains, n_iter
            res[0] = r
        except Exception as e:
            err = e
    raise_if_any_rank_failed(err, device=coll_device, what="pooled summary")
    dist.broadcast_object_list(res, src=0)
    return res[0]


def do_inference_shards(counts, X, exposure_rate, how_many_to_check, *, device=0, coll_device="cpu", chains=None, cores=None,
                        approximate_posterior_analysis=False, lambda_mu_mu=5.612671, adj_prob_theshold=0.05,
                        how_many_posterior_draws=1000, to_exclude=None, truncation_compensation=1.0, seed=1, launch=None,
                        exchange="direct"):
    """One inference pass with the GENES partitioned over the ranks (the reference's map_rect over gene shards,
    inst/stan/negBinomial_MPI.stan:226-240; BASELINE cfg4): every rank holds every world-th gene (the reference's round-robin deal,
    R/utilities.R:125-136: an equal share of the checked genes and of the work on every rank) and runs ALL the chains on them, the six hyper-parameters and the chains' state machines are replicated, and the ranks' partial sums meet every leapfrog
    through the direct exchange (include/ppcx.h ppcx_xchg_*: peer-mapped buffers, no collective call). The checked genes' draws
    are then gathered on rank 0, which computes the credible intervals from them exactly as the chains path does
    (pooled_summary), and the result is broadcast (with the sampler's diagnostics, which are the same on every rank: the state
    machines are replicated). Needs a design that runs pipelined rounds (factor designs, `~ 1`). `launch` = (lanes_per_gene,
    workgroups) pins the log-likelihood launch as in do_inference.
    `exchange` = "direct" (default) or "rccl" (an RCCL all-reduce per leapfrog between the launches of the three-launch round,
    ppcx_fit_nuts_comm; needs one GPU per rank). The direct exchange has run between processes that share one GPU and between
    host threads, never yet between GPUs (no multi-GPU box was available to any round): if its set-up fails on ANY rank -- the
    allocation of uncached memory, an IPC handle that a peer cannot open, peer access -- ALL ranks fall back to the RCCL path
    together, with a warning; a failure of that path is raised on every rank."""
    import math
    import warnings
    from . import _lib
    from .inference import _to_cell_ids, find_optimal_number_of_chains, pooled_summary
    if exchange not in ("direct", "rccl"):
        raise ValueError("exchange must be 'direct' or 'rccl'")
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    counts = np.asarray(counts)
    G, S = counts.shape
    K = int(how_many_to_check)
    X = np.asarray(X, dtype=np.float64).reshape(S, -1)
    C = X.shape[1]
    n2 = max(C - 2, 0)
    practical = 1000 if approximate_posterior_analysis else how_many_posterior_draws
    if chains is None:
        chains = max(3, min(int(cores) if cores else 8, find_optimal_number_of_chains(practical)))
    n_iter = int(math.ceil(practical / chains)) + 150
    n_keep = n_iter - 150
    # genes are dealt to the ranks round-robin, as the reference deals them to its shards (R/utilities.R:125-136): every rank gets
    # its share of the K checked genes -- which come first -- and with them of the slope coordinates and the dearer passes
    mine = np.arange(rank, G, world)
    excl = _to_cell_ids(to_exclude, S)
    if excl is not None and len(excl):
        excl = np.asarray(excl, np.int64)
        eg, es = excl // S, excl % S
        keep = (eg % world) == rank
        excl = ((eg[keep] // world) * S + es[keep]).astype(np.int32)
    part, err, diag = None, None, None
    m = xg = comm = None
    try:
        m = _lib.Model(counts[mine], X, exposure_rate, 0, lambda_mu_mu=lambda_mu_mu, excl=excl, device=device, shard=(G, K, rank, None, world))
    except Exception as e:                          # noqa: BLE001
        err = e
    raise_if_any_rank_failed(err, device=coll_device, what="gene shards (model)")
    if exchange == "direct":
        xerr, handle = None, b""
        try:
            xg = _lib.Xchg(world, rank, chains, device=device)
            handle = xg.handle()
        except Exception as e:                      # noqa: BLE001
            xerr = e
        handles = [None] * world
        dist.all_gather_object(handles, handle)
        if xerr is None:
            try:
                if world > 1:
                    xg.connect(handles)
            except Exception as e:                  # noqa: BLE001
                xerr = e
        if any_rank_failed(xerr, device=coll_device):      # also the barrier before anybody publishes
            if xg is not None:
                xg.close()
                xg = None
            warnings.warn("gene shards: the direct exchange could not be set up on every rank"
                          + (f" (this rank: {xerr})" if xerr is not None else "") + "; all ranks use the RCCL path", RuntimeWarning)
            exchange = "rccl"
    if exchange == "rccl":
        uid = [None]
        try:
            if rank == 0:
                uid[0] = _lib.Comm.unique_id()
        except Exception as e:                      # noqa: BLE001
            err = e
        raise_if_any_rank_failed(err, device=coll_device, what="gene shards (RCCL id)")
        dist.broadcast_object_list(uid, src=0)
        try:
            comm = _lib.Comm(world, rank, uid[0], device=device)
        except Exception as e:                      # noqa: BLE001
            err = e
        raise_if_any_rank_failed(err, device=coll_device, what="gene shards (RCCL communicator)")
    try:
        if launch is not None:
            m.set_launch(*launch)
        if comm is not None:
            f = m.fit_nuts_comm(comm, chains=chains, iter=n_iter, warmup=150, seed=seed)
        else:
            f = m.fit_nuts_xchg(xg, chains=chains, iter=n_iter, warmup=150, seed=seed)
        try:
            diag = f.diagnostics()
            Gl, Kl = len(mine), m.K                 # this shard's genes and checked genes (local unconstrained vector, Stan order)
            a1, a2 = 3 + Gl, 3 + Gl + Kl
            sr = a2 + n2 * Kl
            cols = np.concatenate([np.arange(3), 3 + np.arange(Kl), a1 + np.arange(Kl), a2 + np.arange(n2 * Kl),
                                   sr + np.arange(Kl), sr + Gl + np.arange(3)]).astype(np.int32)
            part = (Kl, f.columns(cols))
        finally:
            f.close()
    except Exception as e:                          # noqa: BLE001
        err = e
    finally:
        if m is not None:
            m.close()
        if xg is not None:
            xg.close()
        if comm is not None:
            comm.close()
    raise_if_any_rank_failed(err, device=coll_device, what="gene shards (fit)")
    parts = [None] * world
    dist.all_gather_object(parts, part)
    res, err = [None], None
    if rank == 0:
        try:
            pooled = np.zeros((chains, n_keep, 3 + K * (2 + max(C - 1, 1)) + 3))
            pooled[..., :3] = parts[0][1][..., :3]                           # hyper-parameters: replicated, rank 0's copy
            pooled[..., -3:] = parts[0][1][..., -3:]
            for r, (Kl, dr) in enumerate(parts):                             # rank r's checked genes are r, r + world, ... of the K
                if Kl:
                    gk = np.arange(r, K, world)                              # their places among the K checked genes
                    pooled[..., 3 + gk] = dr[..., 3:3 + Kl]
                    pooled[..., 3 + K + gk] = dr[..., 3 + Kl:3 + 2 * Kl]
                    for c in range(n2):                                      # alpha_2: (C - 2) entries per checked gene, gene-major
                        pooled[..., 3 + 2 * K + n2 * gk + c] = dr[..., 3 + 2 * Kl + c:3 + 2 * Kl + n2 * Kl:n2]
                    pooled[..., 3 + (2 + n2) * K + gk] = dr[..., 3 + (2 + n2) * Kl:3 + (3 + n2) * Kl]
            r = pooled_summary(counts, X, exposure_rate, K, pooled, lambda_mu_mu=lambda_mu_mu,
                               approximate_posterior_analysis=approximate_posterior_analysis, adj_prob_theshold=adj_prob_theshold,
                               how_many_posterior_draws=how_many_posterior_draws, truncation_compensation=truncation_compensation,
                               seed=seed, device=device)
            r.chains, r.iter = chains, n_iter
            r.diagnostics = diag
            res[0] = r
        except Exception as e:                      # noqa: BLE001
            err = e
    raise_if_any_rank_failed(err, device=coll_device, what="pooled summary")
    dist.broadcast_object_list(res, src=0)
    return res[0]


def identify_outliers(data, *, shards=False, device=0, coll_device="cpu", launch=None, **kw):
    """ppcseq_amd.methods.identify_outliers -- thresholds, discovery pass, exclusion, test pass with truncation compensation,
    flags and the tidy frame (R/methods.R:155-167,268-342) -- with BOTH passes run over the ranks of the initialised
    torch.distributed job, one process per GPU: the chains of a pass dealt to the ranks (do_inference above; the reference's
    chains / cores, R/utilities.R:1500-1501), or with shards=True its genes (do_inference_shards; the reference's map_rect over
    gene shards). Every rank passes the same data and gets the same frame. NUTS only (approximate_posterior_inference = False)."""
    import functools
    from . import methods
    kw.setdefault("approximate_posterior_inference", False)
    one_pass = functools.partial(do_inference_shards if shards else do_inference, device=device, coll_device=coll_device, launch=launch)
    return methods.identify_outliers(data, device=device, _pass=one_pass, **kw)
