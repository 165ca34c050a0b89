"""Multi-GPU paths on whatever the box offers (`-m gpu`):
  * chains partitioned over ranks (one process per GPU; here two ranks share the one GPU over gloo) and over the devices
    of one process: the pooled credible intervals are those of a single fit of the same global chains, bit for bit;
  * gene shards over ranks with the per-leapfrog all-reduce and the rank-divergence guard: over RCCL where two devices are
    visible (one rank per GPU; skipped otherwise), and on ANY box through tests/loopback -- a stand-in for the five nccl*
    entry points over shared memory (bound by the testing build of the library), because RCCL refuses two ranks on one device. The loopback runs
    reproduce the in-process shards bit for bit, and a failure injected into one rank makes both ranks return together.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import independent as ind

pytestmark = pytest.mark.gpu
KW = dict(adj_prob_theshold=0.01, how_many_posterior_draws=1200, truncation_compensation=0.7352941, seed=31)
LAUNCH = (8, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data():
    d = ind.synth(60, 12, K=5, seed=9)
    return d["counts"], d["X"], d["exposure"], d["K"]


def _rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from ppcseq_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        counts, X, expo, K = _data()
        r = D.do_inference(counts, X, expo, K, device=0, coll_device="cpu", chains=4, launch=LAUNCH,
                           to_exclude=np.array([3, 14], np.int32), **KW)
        q.put((rank, r.lower, r.upper, r.mean, r.slope, r.deleterious_outliers, r.chains, r.iter))
    finally:
        dist.destroy_process_group()


def test_chains_over_ranks_and_devices_pool_to_the_single_fit():
    from ppcseq_amd import _lib
    from ppcseq_amd.inference import do_inference
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: the product has no CPU fallback")
    counts, X, expo, K = _data()
    excl = np.array([3, 14], np.int32)
    one = do_inference(counts, X, expo, K, chains=4, launch=LAUNCH, to_exclude=excl, **KW)
    # two devices of one process (here the same device twice: two host threads, two models)
    two = do_inference(counts, X, expo, K, chains=4, launch=LAUNCH, to_exclude=excl, devices=[0, 0], **KW)
    for a, b in [(one.lower, two.lower), (one.upper, two.upper), (one.mean, two.mean), (one.sd, two.sd), (one.slope, two.slope)]:
        assert np.array_equal(a, b)
    assert np.array_equal(one.deleterious_outliers, two.deleterious_outliers) and (two.chains, two.iter) == (one.chains, one.iter)
    # two ranks (one process per GPU; both on the one device here), gloo for the plumbing
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in res:                                  # every rank holds the pooled result
        assert np.array_equal(r[1], one.lower) and np.array_equal(r[2], one.upper) and np.array_equal(r[3], one.mean)
        assert np.array_equal(r[4], one.slope) and np.array_equal(r[5], one.deleterious_outliers)
        assert (r[6], r[7]) == (one.chains, one.iter)


def _shard_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from ppcseq_amd import _lib as L
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = ind.synth(80, 10, K=6, seed=4)
        G, K = 80, d["K"]
        uid = [L.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = L.Comm(world, rank, uid[0], device=rank)
        g0, g1 = G * rank // world, G * (rank + 1) // world
        m = L.Model(d["counts"][g0:g1], d["X"], d["exposure"], 0, device=rank, shard=(G, K, g0, g1))
        f = m.fit_nuts_comm(comm, chains=2, iter=40, warmup=25, seed=6)
        q.put((rank, f.diagnostics()["n_leapfrog"], f.draws()[..., :3]))
        f.close(); m.close(); comm.close()
    finally:
        dist.destroy_process_group()


def test_gene_shards_over_two_ranks_equal_the_unsharded_run():
    from ppcseq_amd import _lib as L
    if L.device_count() < 2:
        pytest.skip("gene shards over RCCL need two devices (one rank per GPU)")
    d = ind.synth(80, 10, K=6, seed=4)
    m = L.Model(d["counts"], d["X"], d["exposure"], d["K"])
    try:
        f = m.fit_nuts(chains=2, iter=40, warmup=25, seed=6)
        nl, hy = f.diagnostics()["n_leapfrog"], f.draws()[..., :3]
        f.close()
    finally:
        m.close()
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in res:
        assert np.array_equal(r[1][:, :12], nl[:, :12])          # same decisions until rounding separates the runs
        assert np.max(np.abs(r[2][:, :3] - hy[:, :3])) < 1e-6    # hyper-parameter draws of the first kept iterations


# ---- gene shards over two ranks on one device, through the loopback collective (tests/loopback/loopback_rccl.cpp) ----
LOOPBACK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback")


def _loopback_lib():
    import subprocess
    lib, src = os.path.join(LOOPBACK_DIR, "libloopback_rccl.so"), os.path.join(LOOPBACK_DIR, "loopback_rccl.cpp")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-fPIC", "-shared",
                               "-fvisibility=hidden", "-o", lib, src, "-lrt"])
    return lib


def _testing_lib():
    from ppcseq_amd import build
    return build.build_testing()


def _loopback_worker(rank, world, conn, q, G, K, kw, hooks):
    """One rank = one process; both ranks on device 0. The communicator's id travels from rank 0 over a pipe. The ranks
    bind the TESTING build of the library: the stand-in nccl provider and the fault injection exist only there."""
    from ppcseq_amd import _lib as L
    try:
        L.use_library(_testing_lib())
        L.testing_set_nccl_provider(_loopback_lib())
        for k, v in (hooks or {}).items():
            L.testing_set(k, v)
        d = ind.synth(G, 10, K=K, seed=4)
        if rank == 0:
            uid = L.Comm.unique_id()
            conn.send(uid)
        else:
            uid = conn.recv()
        comm = L.Comm(world, rank, uid, device=0)
        g0, g1 = G * rank // world, G * (rank + 1) // world
        m = L.Model(d["counts"][g0:g1], d["X"], d["exposure"], 0, device=0, shard=(G, K, g0, g1))
        try:
            f = m.fit_nuts_comm(comm, **kw)
            q.put((rank, "ok", f.diagnostics()["n_leapfrog"], f.draws()))
            f.close()
        except L.PpcxError as e:
            q.put((rank, "error", str(e), None))
        m.close(); comm.close()
    except Exception as e:                      # anything else: reported, so that the parent does not wait for the timeout
        q.put((rank, "crash", repr(e), None))


def _run_two_ranks(G, K, kw, hooks=None):
    _loopback_lib(); _testing_lib()              # built once, here, not by both ranks at the same time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    a, b = ctx.Pipe()
    procs = [ctx.Process(target=_loopback_worker, args=(r, 2, (a, b)[r], q, G, K, kw, hooks)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_gene_shards_over_two_ranks_through_the_loopback_collective():
    """ppcx_fit_nuts_comm with nranks = 2 (the reference's map_rect over gene shards, inst/stan/negBinomial_MPI.stan:226-240,
    one shard per process): the per-round all-reduce between reduce and advance, every rank replicating the state
    machines, the guard's poll. The two ranks' draws are the columns of the in-process two-shard fit, bit for bit (same
    order of the shard sum), and agree with the unsharded fit."""
    from ppcseq_amd import _lib as L
    G, K = 80, 6
    kw = dict(chains=2, iter=40, warmup=25, seed=6)
    d = ind.synth(G, 10, K=K, seed=4)
    shards = [L.Model(d["counts"][g0:g1], d["X"], d["exposure"], 0, shard=(G, K, g0, g1)) for g0, g1 in ((0, 40), (40, 80))]
    whole = L.Model(d["counts"], d["X"], d["exposure"], K)
    try:
        fits = L.fit_nuts_shards(shards, **kw)
        ref = [(f.diagnostics()["n_leapfrog"].copy(), f.draws().copy()) for f in fits]
        for f in fits:
            f.close()
        fw = whole.fit_nuts(**kw)
        nl_w, hy_w = fw.diagnostics()["n_leapfrog"], fw.draws()[..., :3]
        fw.close()
    finally:
        for m in shards + [whole]:
            m.close()
    res = _run_two_ranks(G, K, kw)
    for rank, status, nl, dr in res:
        assert status == "ok", (rank, status, nl)
        assert np.array_equal(nl, ref[rank][0]) and np.array_equal(dr, ref[rank][1])        # = the in-process shards
        assert np.array_equal(nl[:, :12], nl_w[:, :12]) and np.max(np.abs(dr[:, :3, :3] - hy_w[:, :3])) < 1e-6   # ~ unsharded


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_a_failing_rank_takes_its_peer_out_of_the_collectives(failing_rank):
    """Fault injection: one rank fails after round 64 (testing build, ppcx_testing_set). It keeps issuing the per-round all-reduces
    until the poll, where the guard's max-reduction tells both ranks: both return the same error class, neither hangs."""
    kw = dict(chains=2, iter=200, warmup=100, seed=6)
    res = _run_two_ranks(80, 6, kw, {"fail_at_round": 64, "fail_rank": failing_rank})
    assert [r[1] for r in res] == ["error", "error"], res
    assert all("ppcx error -2" in r[2] for r in res), res                     # PPCX_ERR_HIP, the class of the injected failure
    assert "injected failure" in res[failing_rank][2] and "another rank" in res[1 - failing_rank][2]


def _rccl_single_rank_worker(port, q):
    """The collectives of the chains-over-ranks path on device tensors over RCCL (backend "nccl"), a group of one rank: what
    bench.py and distributed.do_inference issue on a GPU node, as far as one GPU can run it."""
    import torch
    import torch.distributed as dist
    from ppcseq_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        counts, X, expo, K = _data()
        got = D.broadcast_arrays(dict(counts=counts, X=X, exposure=expo, K=np.array([K], np.int64)), device="cuda:0")
        ok = all(np.array_equal(got[k], v) and got[k].dtype == v.dtype
                 for k, v in dict(counts=counts, X=X, exposure=expo, K=np.array([K], np.int64)).items())
        x = np.arange(24, dtype=np.float64).reshape(2, 3, 4)
        ok = ok and np.array_equal(D.all_gather_chains(x, device="cuda:0"), x) and D.max_over_ranks(2.5, device="cuda:0") == 2.5
        r = D.do_inference(counts, X, expo, K, device=0, coll_device="cuda:0", chains=3, launch=LAUNCH, **KW)
        q.put(("ok" if ok else "mismatch", r.lower, r.upper, r.deleterious_outliers))
    except Exception as e:
        q.put(("crash: " + repr(e), None, None, None))
    finally:
        dist.destroy_process_group()


def test_collectives_over_rccl_with_one_rank():
    from ppcseq_amd.inference import do_inference
    counts, X, expo, K = _data()
    one = do_inference(counts, X, expo, K, chains=3, launch=LAUNCH, **KW)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), q))
    p.start()
    status, lower, upper, flags = q.get(timeout=600)
    p.join(120)
    assert status == "ok" and p.exitcode == 0, status
    assert np.array_equal(lower, one.lower) and np.array_equal(upper, one.upper) and np.array_equal(flags, one.deleterious_outliers)
