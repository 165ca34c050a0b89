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


# ---- gene shards with the DIRECT exchange (include/ppcx.h ppcx_xchg_*): the ranks' state machines add their partial sums
# ---- themselves through peer-mapped buffers, inside the merged launch of a pipelined round ------------------------------
XG, XK, XS = 80, 6, 10
XKW = dict(chains=2, iter=40, warmup=25, seed=6)


def _xchg_reference():
    """The unsharded fit and the in-process shards over the three-launch round, for the decisions of the first iterations."""
    from ppcseq_amd import _lib as L
    d = ind.synth(XG, XS, K=XK, seed=4)
    whole = L.Model(d["counts"], d["X"], d["exposure"], XK)
    shards = [L.Model(d["counts"][g0:g1], d["X"], d["exposure"], 0, shard=(XG, XK, g0, g1)) for g0, g1 in ((0, 40), (40, 80))]
    try:
        fw = whole.fit_nuts(**XKW)
        nl_w, ss_w, hy_w = fw.diagnostics()["n_leapfrog"].copy(), fw.diagnostics()["stepsize"].copy(), fw.draws()[..., :3].copy()
        fw.close()
        fits = L.fit_nuts_shards(shards, **XKW)
        nl_s = fits[0].diagnostics()["n_leapfrog"].copy()
        for f in fits:
            f.close()
    finally:
        for m in shards + [whole]:
            m.close()
    return d, nl_w, ss_w, hy_w, nl_s


def _check_ranks(res, nl_w, ss_w, hy_w, nl_s):
    (nl0, ss0, dr0), (nl1, ss1, dr1) = res
    assert np.array_equal(nl0, nl1) and np.array_equal(ss0, ss1)                 # replicated state machines: identical decisions
    assert np.array_equal(dr0[..., :3], dr1[..., :3]) and np.array_equal(dr0[..., -3:], dr1[..., -3:])   # hyper draws, bit for bit
    assert np.array_equal(nl0[:, :12], nl_w[:, :12]) and np.array_equal(nl0[:, :12], nl_s[:, :12])       # = unsharded = three-launch shards
    assert np.max(np.abs(ss0[:, :12] - ss_w[:, :12])) < 1e-9
    assert np.max(np.abs(dr0[:, :3, :3] - hy_w[:, :3])) < 1e-6


def test_direct_exchange_between_two_ranks_in_one_process():
    """Two shard models, two host threads, one exchange group wired with plain device pointers (ppcx_xchg_connect_local)."""
    import threading
    from ppcseq_amd import _lib as L
    d, nl_w, ss_w, hy_w, nl_s = _xchg_reference()
    shards = [L.Model(d["counts"][g0:g1], d["X"], d["exposure"], 0, shard=(XG, XK, g0, g1)) for g0, g1 in ((0, 40), (40, 80))]
    xs = L.Xchg.local_group(2, XKW["chains"])
    res, err = [None, None], [None, None]

    def run(k):
        try:
            xs[k].set_timeout(30)
            f = shards[k].fit_nuts_xchg(xs[k], **XKW)
            dg = f.diagnostics()
            res[k] = (dg["n_leapfrog"], dg["stepsize"], f.draws())
            err[k] = f.xchg_timing()
            f.close()
        except Exception as e:                   # noqa: BLE001
            err[k] = e
    try:
        th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert all(r is not None for r in res), err
        _check_ranks(res, nl_w, ss_w, hy_w, nl_s)
        assert err[0][1] > 0 and err[0][1] == err[1][1]                          # both ranks took part in the same exchanges
    finally:
        for x in xs:
            x.close()
        for m in shards:
            m.close()


def _xchg_worker(rank, conn, q, hooks):
    """One rank = one process, both on device 0; the 64-byte IPC handles travel over a pipe."""
    from ppcseq_amd import _lib as L
    try:
        if hooks:
            from ppcseq_amd import build
            L.use_library(build.build_testing())
            for k, v in hooks.items():
                L.testing_set(k, v)
        d = ind.synth(XG, XS, K=XK, seed=4)
        # the round-robin deal of the reference (R/utilities.R:125-136): rank r holds genes r, r + 2, ... -- three of the six
        # checked genes each (the in-process tests above split the genes contiguously: both forms run)
        m = L.Model(d["counts"][rank::2], d["X"], d["exposure"], 0, device=0, shard=(XG, XK, rank, None, 2))
        x = L.Xchg(2, rank, XKW["chains"], device=0)
        x.set_timeout(30)
        mine = x.handle()
        conn.send(mine)
        theirs = conn.recv()
        x.connect([mine, theirs] if rank == 0 else [theirs, mine])
        conn.send(b"connected"); conn.recv()          # nobody publishes before both have mapped the other's buffer
        try:
            kw = dict(XKW, iter=200, warmup=100) if hooks else XKW
            f = m.fit_nuts_xchg(x, **kw)
            dg = f.diagnostics()
            q.put((rank, "ok", (dg["n_leapfrog"], dg["stepsize"], f.draws()), f.xchg_timing()))
            f.close()
        except L.PpcxError as e:
            q.put((rank, "error", str(e), None))
        m.close(); x.close()
    except Exception as e:                       # noqa: BLE001
        q.put((rank, "crash", repr(e), None))


def _run_xchg_ranks(hooks=None):
    if hooks:
        from ppcseq_amd import build
        build.build_testing()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    a, b = ctx.Pipe()
    procs = [ctx.Process(target=_xchg_worker, args=(r, (a, b)[r], q, hooks)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def test_direct_exchange_between_two_processes_over_ipc_handles():
    """ppcx_fit_nuts_xchg with one rank per PROCESS (the reference's map_rect over gene shards, .stan:226-240, one shard per
    GPU): the receive buffers are mapped through hipIpc handles -- which also works between two processes on ONE device, so
    the real exchange path runs on a one-GPU box. Same decisions and hyper draws on both ranks, those of the unsharded fit."""
    d, nl_w, ss_w, hy_w, nl_s = _xchg_reference()
    res = _run_xchg_ranks()
    assert [r[1] for r in res] == ["ok", "ok"], res
    _check_ranks([r[2] for r in res], nl_w, ss_w, hy_w, nl_s)
    assert res[0][3][1] > 0 and res[0][3][1] == res[1][3][1]


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_a_rank_that_fails_takes_its_peer_out_of_the_direct_exchange(failing_rank):
    """Fault injection (testing build): one rank fails after round 64. It tells its peer (abort word in the peer's buffer);
    the peer's state machines stop waiting and its fit ends with PPCX_ERR_STALL -- neither rank hangs."""
    res = _run_xchg_ranks({"fail_at_round": 64, "fail_rank": failing_rank})
    assert [r[1] for r in res] == ["error", "error"], res
    assert "injected failure" in res[failing_rank][2]
    assert "ppcx error -5" in res[1 - failing_rank][2] and "peer rank left" in res[1 - failing_rank][2], res


def _shards_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from ppcseq_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d = ind.synth_factor(90, 12, 7, (3,), 5)
        r = D.do_inference_shards(d["counts"], d["X"], d["exposure"], 7, device=0, coll_device="cpu", chains=4,
                                  to_exclude=np.array([3, 14], np.int32), launch=LAUNCH, **KW)
        q.put((rank, "ok", r.lower, r.upper, r.slope, r.deleterious_outliers, r.diagnostics["n_leapfrog"], r.diagnostics["divergent"]))
    except Exception as e:                      # noqa: BLE001
        q.put((rank, "crash: " + repr(e), None, None, None, None, None, None))
    finally:
        dist.destroy_process_group()


def test_do_inference_with_the_genes_sharded_over_two_ranks():
    """distributed.do_inference_shards: genes over the ranks (here two gloo ranks on the one GPU, a three-level factor design:
    C = 3), all chains on every rank, the direct exchange every leapfrog, the checked genes' draws gathered on rank 0 for the
    posterior-predictive pass. Against the same pass on one rank with all the genes: identical decisions for as long as rounding
    lets the chains coincide, so the intervals agree within Monte-Carlo error and the slopes closely."""
    from ppcseq_amd.inference import do_inference
    d = ind.synth_factor(90, 12, 7, (3,), 5)
    one = do_inference(d["counts"], d["X"], d["exposure"], 7, chains=4, to_exclude=np.array([3, 14], np.int32), launch=LAUNCH, **KW)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shards_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["ok", "ok"], [r[1] for r in res]
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][5], res[1][5])          # broadcast: the same on every rank
    lower, upper, slope, nl, div = res[0][2], res[0][3], res[0][4], res[0][6], res[0][7]
    # the same launch geometry (lanes per gene pinned) and the same Philox streams: the sharded chains take the unsharded chains'
    # decisions for the first iterations, until the different summation order of the shards' sums has grown into the trees
    assert np.array_equal(nl[:, :10], one.diagnostics["n_leapfrog"][:, :10])
    # ... and afterwards they are other draws of the same posterior: slopes and interval ends agree within Monte-Carlo error.
    # The Monte-Carlo standard error of a posterior mean over n = 1200 draws of autocorrelated chains is about sd / sqrt(n / 4);
    # the 1 % / 99 % ends of a predictive interval from 1200 draws move by several per cent of their value between runs.
    nd = one.diagnostics["n_leapfrog"].shape[0] * (one.diagnostics["n_leapfrog"].shape[1] - 150)
    assert np.max(np.abs(slope - one.slope)) < 6 * 0.35 / np.sqrt(nd / 4)          # posterior sd of a slope here: 0.2-0.35
    assert np.median(np.abs(upper - one.upper) / (1 + one.upper)) < 0.05 and np.max(np.abs(upper - one.upper) / (1 + one.upper)) < 0.3
    assert np.median(np.abs(lower - one.lower) / (1 + one.lower)) < 0.08
    assert np.array_equal(res[0][5], one.deleterious_outliers)                                    # the same calls
    # divergent transitions after warm-up are a property of this small problem (12 samples per gene, a three-level factor), not of
    # the sharding: the unsharded fit and the oracle's sampler meet them at the same seed as well
    # (tests/test_oracle_nuts.py::test_small_factor_problem_diverges_on_the_oracle_too)
    assert int(div[:, 150:].sum()) <= 40 and int(one.diagnostics["divergent"][:, 150:].sum()) <= 40


def _shards_fallback_worker(rank, world, port, q):
    import warnings
    import torch.distributed as dist
    from ppcseq_amd import _lib as L, distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L.use_library(_testing_lib())
        L.testing_set_nccl_provider(_loopback_lib())      # RCCL refuses two ranks on one device: the stand-in over shared memory
        if rank == 1:                                     # this rank cannot map its peer's buffer
            def broken_connect(self, handles):
                raise L.PpcxError("ppcx error -3: injected: hipIpcOpenMemHandle failed")
            L.Xchg.connect = broken_connect
        d = ind.synth_factor(90, 12, 7, (3,), 5)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            r = D.do_inference_shards(d["counts"], d["X"], d["exposure"], 7, device=0, coll_device="cpu", chains=4,
                                      to_exclude=np.array([3, 14], np.int32), launch=LAUNCH, **KW)
        q.put((rank, "ok", r.deleterious_outliers, r.diagnostics["n_leapfrog"], [str(x.message) for x in w]))
    except Exception as e:                      # noqa: BLE001
        q.put((rank, "crash: " + repr(e), None, None, None))
    finally:
        dist.destroy_process_group()


def test_sharded_do_inference_falls_back_to_the_collective_when_the_direct_exchange_cannot_be_set_up():
    """The direct exchange has never run between GPUs (no multi-GPU box was available to any round): when its set-up fails on
    ANY rank -- here rank 1 cannot map its peer's buffer -- ALL ranks of distributed.do_inference_shards take the RCCL path
    together (ppcx_fit_nuts_comm; through the loopback stand-in here, two ranks on one device), warn, and deliver the calls of the
    one-rank pass; nobody is left waiting in a collective."""
    from ppcseq_amd.inference import do_inference
    _loopback_lib(); _testing_lib()
    d = ind.synth_factor(90, 12, 7, (3,), 5)
    one = do_inference(d["counts"], d["X"], d["exposure"], 7, chains=4, to_exclude=np.array([3, 14], np.int32), launch=LAUNCH, **KW)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shards_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["ok", "ok"], [r[1] for r in res]
    for r in res:
        assert any("all ranks use the RCCL path" in m for m in r[4]), r[4]
    assert "injected" in " ".join(res[1][4]) and "injected" not in " ".join(res[0][4])          # the failing rank says why
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][2], one.deleterious_outliers)
    assert np.array_equal(res[0][3][:, :10], one.diagnostics["n_leapfrog"][:, :10])              # the unsharded chains' first trees


def test_a_refused_exchange_fit_leaves_nothing_on_the_device():
    """ppcx_fit_nuts_xchg looks at its exchange group before it allocates anything: the direct exchange needs pipelined rounds,
    which ppcx_model_set_rounds can rule out -- a documented, expected refusal (PPCX_ERR_LIMIT: use ppcx_fit_nuts_comm) -- and a
    refused call must leave no draws buffer behind and no reference that would keep the model's count matrix on the device after
    Model.close()."""
    from ppcseq_amd import _lib as L
    if L.device_count() < 1:
        pytest.fail("no HIP device visible: the product has no CPU fallback")
    d = ind.synth(300, 40, K=20, seed=31, C=2)
    X = d["X"].copy()
    free0, _ = L.device_memory(0)
    m = L.Model(d["counts"], X, d["exposure"], 20)
    m.set_rounds(pipelined=0)                                               # the three-launch round: no direct exchange
    xg = L.Xchg(1, 0, 4)
    try:
        for _ in range(3):
            with pytest.raises(L.PpcxError, match="ppcx error -6"):          # PPCX_ERR_LIMIT
                m.fit_nuts_xchg(xg, chains=4, iter=2000, warmup=100, seed=1)   # would hold 4 x 1900 x D doubles of draws
        free1, _ = L.device_memory(0)
    finally:
        xg.close(); m.close()
    free2, _ = L.device_memory(0)
    assert free0 - free1 < 64 << 20, "the refused fits left their draws on the device"   # the model itself is a few MB
    assert abs(free2 - free0) < 16 << 20, "Model.close() was deferred by a fit that was never handed out"


# ---- identify_outliers() -- both passes -- over several devices and over several ranks (R/methods.R:268-342 with the reference's
# ---- chains spread over `cores`, R/utilities.R:1500-1501) -------------------------------------------------------------------
def _two_pass_frame():
    """A reduced cfg5: synthetic genes with injected outliers, two groups, pfp = 5, the first K genes checked."""
    import pandas as pd
    from ppcseq_amd.synth import synth
    d = synth(400, 24, seed=20255)
    K = 20
    G, S = d["counts"].shape
    rows = []
    for g in range(G):
        for s_ in range(S):
            rows.append((f"s{s_:02d}", f"g{g:04d}", int(d["counts"][g, s_]), "B" if d["X"][s_, 1] else "A", g < K, 0.5 if g < K else 0.9 + 1e-6 * g))
    return pd.DataFrame(rows, columns=["sample", "symbol", "value", "Label", "is_significant", "PValue"]), K


_TWO_PASS_KW = dict(formula="~ Label", sample="sample", transcript="symbol", abundance="value", significance="PValue", do_check="is_significant",
                    percent_false_positive_genes=5, how_many_negative_controls=380, approximate_posterior_inference=False,
                    approximate_posterior_analysis=False, cores=4, seed=77, launch=LAUNCH)


def _summary(out):
    sw = out["sample_wise_data"]
    return (out["ppc_samples_failed"].tolist(), out["tot_deleterious_outliers"].tolist(),
            np.stack([f[".upper"].to_numpy() for f in sw]), np.stack([f["slope_after_outlier_filtering"].to_numpy()[0:1] for f in sw]))


def _two_pass_rank_worker(rank, world, port, q):
    import torch.distributed as dist
    from ppcseq_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data, _ = _two_pass_frame()
        kw = {k: v for k, v in _TWO_PASS_KW.items() if k != "launch"}
        out = D.identify_outliers(data, device=0, coll_device="cpu", launch=LAUNCH, **kw)
        q.put((rank, "ok") + _summary(out))
    except Exception as e:                      # noqa: BLE001
        q.put((rank, "crash: " + repr(e), None, None, None, None))
    finally:
        dist.destroy_process_group()


def test_identify_outliers_over_two_devices_and_over_two_ranks_equals_the_single_device():
    """Both passes of identify_outliers with the chains dealt (a) to two devices of this process (devices=[0, 0]: two host
    threads) and (b) to two ranks of a torch.distributed job (gloo here, one GPU shared): global chain ids and pinned lanes per
    gene make the pooled chains those of the one-device run, so the frames coincide -- the same flagged cells, the same interval
    ends and slopes, bit for bit."""
    from ppcseq_amd import _lib
    from ppcseq_amd.methods import identify_outliers
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: the product has no CPU fallback")
    data, K = _two_pass_frame()
    one = _summary(identify_outliers(data, device=0, **_TWO_PASS_KW))
    two = _summary(identify_outliers(data, devices=[0, 0], **_TWO_PASS_KW))
    assert sum(one[1]) >= 1                                                 # the injected outliers are found
    assert one[0] == two[0] and one[1] == two[1] and np.array_equal(one[2], two[2]) and np.array_equal(one[3], two[3])
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_pass_rank_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert [r[1] for r in res] == ["ok", "ok"], [r[1] for r in res]
    for r in res:
        assert r[2] == one[0] and r[3] == one[1] and np.array_equal(r[4], one[2]) and np.array_equal(r[5], one[3])
