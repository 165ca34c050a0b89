"""Multi-GPU layer: one process per GPU, chains partitioned across ranks.

Mirrors the reference's chain parallelism (`chains=`/`cores=` of rstan::sampling, R/utilities.R:1500-1501,
which forks one R worker per chain): chains are independent units, so there is NO per-leapfrog
collective. torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests)
is used for plumbing only: rank 0 broadcasts the model inputs, and the kept hyper-parameter draws /
diagnostics are all-gathered for the pooled summaries.
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
