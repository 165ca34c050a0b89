"""world_size-2 gloo test of the multi-GPU plumbing (chains partitioned over ranks, no data-path
collective): every global chain id is owned by exactly one rank, inputs broadcast from rank 0 arrive
intact, and the all-gathered draws are in global chain order."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from ppcseq_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chains_per_rank = 3
        off = D.chain_id_offset(rank, chains_per_rank)
        arrays = None
        if rank == 0:
            arrays = dict(counts=np.arange(12, dtype=np.int32).reshape(3, 4), X=np.ones((4, 2)), exposure=np.linspace(-1, 1, 4))
        got = D.broadcast_arrays(arrays)
        # stand-in for a fit: "draws" that encode the global chain id
        local = np.stack([np.full((5, 2), float(off + c)) for c in range(chains_per_rank)])
        allc = D.all_gather_chains(local)
        t = D.max_over_ranks(1.0 + rank)
        q.put((rank, off, got["counts"].tolist(), got["exposure"].tolist(), allc[:, 0, 0].tolist(), t))
    finally:
        dist.destroy_process_group()


def test_chain_partition_broadcast_and_gather():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [0, 3]
    for r in res:
        assert r[2] == np.arange(12).reshape(3, 4).tolist()
        assert r[3] == np.linspace(-1, 1, 4).tolist()
        assert r[4] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0]          # global chain order
        assert r[5] == 2.0


def _failing_worker(rank, world, port, q):
    import torch.distributed as dist
    from ppcseq_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        try:
            D.raise_if_any_rank_failed(None, what="round 1")                # nobody failed: nobody raises
            D.raise_if_any_rank_failed(MemoryError("out of device memory") if rank == 1 else None, what="round 2")
            q.put((rank, "no exception"))
        except Exception as e:
            q.put((rank, type(e).__name__ + ": " + str(e)))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_is_seen_by_every_rank_before_the_gather():
    """distributed.do_inference checks for failures collectively before its all-gather: when one rank's fit raises (out of
    memory, no finite initial point), the others must leave with it instead of waiting in the gather for ever."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[1] == "MemoryError: out of device memory"
    assert res[0].startswith("RuntimeError: round 2: rank 1 failed")
