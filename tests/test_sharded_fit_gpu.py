"""Multi-rank runs through the communicator INSIDE libpmf_hip.so (pmf_comm_init / pmf_hip.dist.Comm).

The test box has one GPU and RCCL refuses two ranks on one device, so the multi-rank cases run
over the library's rehearsal transport (`hostshm`: same chunk pipeline, same event ordering between
the compute and the collective stream, the all-reduce itself done through shared memory), and the
real RCCL call sequence runs with one rank.  The worker processes never import torch."""
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from helpers import gamma_stats, gauss_stats

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _data():
    sys.path[:0] = [os.path.join(ROOT, "prob-matrix-factorization_amd")]
    from pmf_hip.synth import synth_ratings, train_val_split
    u, i, r = synth_ratings(3000, 400, 60000, seed=21)
    u[0], i[0] = 2999, 399
    (tu, ti, tr), (vu, vi, vr) = train_val_split(u, i, r)
    tu[0], ti[0] = 2999, 399
    train = pd.DataFrame({"u": tu, "i": ti, "rating": tr})
    val = pd.DataFrame({"u": np.append(vu, 3005), "i": np.append(vi, 7), "rating": np.append(vr, 4.0)})
    return train, val


def _build(kind, comm=None, presharded=False):
    if presharded:
        model, keys = _build(kind, comm)
        return type(model)(model.config, dtype="f64", comm=comm, presharded=True), keys
    if kind == "hpf":
        from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
        return HPF_CAVI(HPF_CAVI_Config(n_factors=12, a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0,
                                        max_iter=12, tol=1e-3, verbose=False), dtype="f64", comm=comm), \
            ("E_theta", "E_beta", "E_xi", "E_eta", "gamma_a_theta", "gamma_b_beta")
    if kind == "poisson":
        from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
        return PoissonMFCAVI(PoissonMFCAVIConfig(n_factors=12, a0=0.1, b0=0.5, max_iter=6, tol=None, verbose=False),
                             dtype="f64", comm=comm), ("E_theta", "E_beta", "a_theta", "b_beta")
    if kind == "sgd":
        from src.models.gaussian_mf_sgd import GaussianMFSGD, GaussianMFSGDConfig
        return GaussianMFSGD(GaussianMFSGDConfig(n_factors=12, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0,
                                                 lr=0.01, max_iter=4, tol=-1e9, verbose=False), dtype="f64", comm=comm), \
            ("m_theta", "m_beta", "m_user_bias", "m_item_bias")
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    return GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=12, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0,
                                               max_iter=5, tol=-1.0, verbose=False), dtype="f64", comm=comm), \
        ("m_theta", "m_beta", "m_user_bias", "m_item_bias")


def _fit(kind, model, train, val, gm=None):
    if kind in ("gauss", "sgd"):
        gm = float(train["rating"].mean()) if gm is None else gm     # (presharded ranks are given the global mean)
        a, b = train.copy(), val.copy()
        a["rating"] -= gm; b["rating"] -= gm
        model.fit(a, val_df=b, global_mean=gm)
        return model.predict(val["u"].to_numpy(), val["i"].to_numpy(), gm)
    if kind == "hpf":
        a, b = train.copy(), val.copy()
        a["rating"] += 1; b["rating"] += 1
        model.fit(a, val_df=b)
    else:
        model.fit(train, val_df=val)
    return model.predict(val["u"].to_numpy(), val["i"].to_numpy())


def _worker(rank, world, port, kind, out_dir, exchange=None):
    """One rank of a sharded fit: a fresh interpreter (spawn) that must get by without torch.
    `exchange`: None = the library's choice (reduce-scatter -> finalize owned rows -> all-gather for the Gaussian
    factor sweep, all-reduce for the rest), or 'allreduce' / 'scatter_gather' for every item half-sweep."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PMF_COMM_TRANSPORT="hostshm")
    os.environ["PMF_DIST_CHUNKS"] = "3"   # the messages here are too small for the default to pipeline
    from pmf_hip import dist as pdist
    comm = pdist.init_from_env(device=0, exchange=exchange)
    assert comm.world == world and comm.rank == rank and comm.transport == "hostshm"
    train, val = _data()
    pre = kind.endswith("_pre")
    kind = kind[:-4] if pre else kind
    gm_all = float(train["rating"].mean())
    if pre:
        # this rank is handed ONLY its own rows (global user ids): no rank holds the full frames
        b = pdist.shard_bounds(train["u"].to_numpy(), int(train["u"].max()) + 1, world)
        lo, hi = int(b[rank]), int(b[rank + 1])
        full_val = val
        train = train[(train["u"] >= lo) & (train["u"] < hi)]
        val = val[(val["u"] >= lo) & ((val["u"] < hi) | (rank == world - 1))]
    model, keys = _build(kind, comm=comm, presharded=pre)
    pred = _fit(kind, model, train, val, gm_all)
    if pre:   # predict works on every rank for ALL pairs afterwards
        pred = (model.predict(full_val["u"].to_numpy(), full_val["i"].to_numpy(), gm_all) if kind in ("gauss", "sgd")
                else model.predict(full_val["u"].to_numpy(), full_val["i"].to_numpy()))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pred=pred, val_rmse=np.array(model.history_["val_rmse"]),
             iters=model.history_["iterations"], **{k: getattr(model, k) for k in keys})
    if kind == "gauss":
        lo, hi = model.user_range
        local = model.V_theta                       # this rank's rows only: no hidden collective
        assert local.shape[0] == hi - lo
        full = model.gather_V_theta()               # explicit collective, every rank calls it
        assert np.array_equal(full[lo:hi], local)
        np.save(os.path.join(out_dir, f"V{rank}.npy"), full[::97])
    assert "torch" not in sys.modules, "the sharded CAVI path must not import torch"
    comm.barrier()
    model.close()
    comm.close()


def _spawn(kind, out_dir, world=2, exchange=None):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, out_dir, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
    for p in procs:
        if p.is_alive():
            p.kill()
            p.join()
            raise AssertionError("a rank hung")
    assert [p.exitcode for p in procs] == [0] * world


@pytest.mark.parametrize("kind,exchange", [("hpf", None), ("poisson", None), ("gauss", None), ("hpf_pre", None),
                                           ("gauss_pre", None), ("hpf", "scatter_gather"), ("poisson", "scatter_gather"),
                                           ("gauss", "allreduce")])
def test_two_rank_fit_matches_single_process(kind, exchange, tmp_path):
    """`*_pre`: the presharded form -- every rank is given only its own training / validation rows.  `exchange`:
    None = the library's choice per sweep (Gaussian factor sweep: reduce-scatter -> each rank solves its half of
    every chunk's items -> all-gather; the others: all-reduce), or one exchange forced on every item half-sweep."""
    train, val = _data()
    model, keys = _build(kind[:-4] if kind.endswith("_pre") else kind)
    pred = _fit(kind[:-4] if kind.endswith("_pre") else kind, model, train, val)
    _spawn(kind, str(tmp_path), exchange=exchange)
    kind = kind[:-4] if kind.endswith("_pre") else kind
    for rank in range(2):
        d = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(d["iters"]) == model.history_["iterations"]
        np.testing.assert_allclose(d["val_rmse"], model.history_["val_rmse"], rtol=1e-10)
        for k in keys:
            np.testing.assert_allclose(d[k], getattr(model, k), rtol=1e-9, atol=1e-11, err_msg=k)
        np.testing.assert_allclose(d["pred"], pred, rtol=1e-9, atol=1e-11)
        if kind == "gauss":
            np.testing.assert_allclose(np.load(os.path.join(tmp_path, f"V{rank}.npy")), model.V_theta[::97],
                                       rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("kind,world,exchange", [("hpf", 3, None), ("gauss_pre", 4, None), ("gauss", 3, None),
                                                 ("hpf", 3, "scatter_gather")])
def test_more_ranks_than_two_match_single_process(kind, world, exchange, tmp_path):
    """Nothing in the path is specific to two ranks: 3 and 4 ranks on the one GPU (hostshm), full-frame and
    presharded, against the single-process fit.  (400 items in 3 chunks over 3 ranks: sub-ranges of 44 rows and
    one row per chunk left over for the all-reduce of the scatter-gather exchange.)"""
    base = kind[:-4] if kind.endswith("_pre") else kind
    train, val = _data()
    model, keys = _build(base)
    pred = _fit(base, model, train, val)
    _spawn(kind, str(tmp_path), world=world, exchange=exchange)
    for rank in range(world):
        d = np.load(os.path.join(tmp_path, f"rank{rank}.npz"))
        assert int(d["iters"]) == model.history_["iterations"]
        np.testing.assert_allclose(d["val_rmse"], model.history_["val_rmse"], rtol=1e-10)
        for k in keys:
            np.testing.assert_allclose(d[k], getattr(model, k), rtol=1e-9, atol=1e-11, err_msg=k)
        np.testing.assert_allclose(d["pred"], pred, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("exchange", [None, "scatter_gather"])
def test_two_rank_gradient_mode_fit_is_consistent(exchange, tmp_path):
    """The gradient mode averages the displacements of an item's pieces, so a sharded fit is a
    different (equally valid) trajectory than the single-process one: the ranks must agree with
    each other exactly and follow the same validation curve within 3 %."""
    train, val = _data()
    model, keys = _build("sgd")
    _fit("sgd", model, train, val)
    _spawn("sgd", str(tmp_path), exchange=exchange)
    d0, d1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    for k in keys + ("pred", "val_rmse"):
        assert np.array_equal(d0[k], d1[k]), k
    assert int(d0["iters"]) == model.history_["iterations"] == 4
    np.testing.assert_allclose(d0["val_rmse"], model.history_["val_rmse"], rtol=3e-2)


def _bench(extra, world=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1", "--small",
           "--transport", "hostshm", "--share-gpu", "--only"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("workload,chunks,exchange", [("gaussian_mf", 4, "auto"), ("hpf_cavi", 4, "auto"), ("gaussian_mf", 1, "auto"),
                                                      ("gaussian_mf_sgd", 4, "auto"), ("gaussian_mf", 4, "allreduce"),
                                                      ("hpf_cavi", 4, "scatter_gather")])
def test_bench_two_rank_rehearsal_strong_scales_the_one_matrix(workload, chunks, exchange):
    """bench.py launched as the driver launches it (torch.distributed.run, 2 ranks; the hostshm transport
    and one shared GPU stand in for RCCL over two): the ONE matrix is sharded (strong scaling), the
    replicated item state ends bit-identical on both ranks -- under either exchange of the item statistics --
    and the ranks never import torch."""
    res = _bench(["--workload", workload, "--chunks", str(chunks), "--exchange", exchange])
    assert res["config"]["exchange"] == exchange
    assert res["n_gpus"] == 2 and res["scaling"] == "strong"
    assert res["config"]["ratings_total"] == 5_000_000 and res["config"]["n_users"] == 100_000
    assert 0 < res["config"]["ratings_on_rank0"] < 5_000_000
    assert res["config"]["item_replicas_identical"] is True
    assert res["config"]["torch_imported"] is False
    assert res["comm_exposed_ms"] >= 0 and res["comm_allreduce_ms"] > 0


def test_bench_two_rank_rehearsal_weak_scaling_flag():
    res = _bench(["--workload", "hpf_cavi", "--scaling", "weak"])
    assert res["scaling"] == "weak" and res["config"]["ratings_total"] == 10_000_000
    assert res["config"]["item_replicas_identical"] is True


def test_single_gpu_bench_line_is_complete_and_torch_free():
    env = dict(os.environ)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--small", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["config"]["torch_imported"] is False
    # (--small: the 83 MB item covariance table sits in the Infinity Cache, so the HBM fraction is not
    # bounded by 1 here as it is at the full size; only its presence is checked)
    assert res["roofline"]["bound"] == "hbm" and res["roofline"]["frac"] > 0
    hpf = res["also"]["hpf_cavi"]["roofline"]
    assert hpf["bound"] == "cache_gather" and 0 < hpf["frac"] <= 1.0, hpf
    assert set(res["also"]["f64"]) == {"gaussian_mf", "hpf_cavi"}


def test_pipelined_item_sweep_over_real_rccl_single_rank():
    """The one-GPU box cannot hold two RCCL ranks, but it can run the real thing with one: a context with
    a one-rank RCCL communicator takes the in-library three-stage path (accumulate, ncclAllReduce of each
    chunk's slice on the collective stream -- or ncclReduceScatter, finalize of the owned rows, ncclAllGather of the
    finalised state --, event-ordered finalize) and must give bit for bit what the explicit accumulate / finalize
    calls give on a caller-owned buffer without any collective."""
    import torch
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER, TRANSPORT_RCCL, dist as pdist
    from pmf_hip.engine import Context
    from pmf_hip.synth import synth_ratings

    dev = torch.device("cuda", 0)
    comm = pdist.Comm(0, 1, 0, Context.comm_unique_id(), "rccl")
    try:
        U, I, N, K = 20000, 3000, 600000, 64
        u, i, r = synth_ratings(U, I, N, seed=4)
        rng = np.random.default_rng(0)
        m_u, m_i = 0.1 * rng.standard_normal((U, K)), 0.1 * rng.standard_normal((I, K))
        out = []
        for with_comm in ("allreduce", "scatter_gather", False):   # ncclAllReduce | ncclReduceScatter + ncclAllGather | no collective
            ctx = pmf_hip.Context(U, I, K)
            if with_comm:
                comm.attach(ctx)
                ctx.comm_set_exchange(with_comm)
                assert ctx.comm_info() == (1, 0, TRANSPORT_RCCL)
            ctx.set_row_chunks(ITEM, 4)
            ctx.set_ratings(u, i, r - r.mean())
            ctx.set_array(USER, ARR_FACTOR, m_u); ctx.set_array(ITEM, ARR_FACTOR, m_i)
            ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
            ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))
            s_item, s_bias = gauss_stats(ctx, dev)
            for _ in range(3):
                ctx.gauss_factor_sweep(USER, 0.5, 1.0)
                if with_comm:
                    ctx.gauss_factor_sweep(ITEM, 0.5, 1.0)
                else:
                    for c in range(4):
                        ctx.select_chunk(ITEM, c); ctx.gauss_factor_accumulate(ITEM, s_item.ptr)
                    for c in range(4):
                        ctx.select_chunk(ITEM, c); ctx.gauss_factor_finalize(ITEM, s_item.ptr, 0.5, 1.0)
                    ctx.select_chunk(ITEM, -1)
                ctx.gauss_bias_sweep(USER, 0.5, 1.0)
                if with_comm:
                    ctx.gauss_bias_sweep(ITEM, 0.5, 1.0)
                else:
                    ctx.gauss_bias_accumulate(ITEM, s_bias.ptr); ctx.gauss_bias_finalize(ITEM, s_bias.ptr, 0.5, 1.0)
            out.append([ctx.get_array(s, a) for s in (USER, ITEM) for a in (ARR_FACTOR, ARR_COV, ARR_BIAS)])
            g_item = gamma_stats(ctx, dev)
            ctx.set_array(USER, ARR_FACTOR, np.abs(m_u) + 0.1); ctx.set_array(ITEM, ARR_FACTOR, np.abs(m_i) + 0.1)
            ctx.set_ratings(u, i, r + 1.0)
            for _ in range(3):
                ctx.gamma_sweep(USER, 0.3, 0.3)
                if with_comm:
                    ctx.gamma_sweep(ITEM, 0.3, 0.3)
                else:
                    for c in range(4):
                        ctx.select_chunk(ITEM, c); ctx.gamma_accumulate(ITEM, g_item.ptr)
                    for c in range(4):
                        ctx.select_chunk(ITEM, c); ctx.gamma_finalize(ITEM, g_item.ptr, 0.3, 0.3)
                    ctx.select_chunk(ITEM, -1)
            out[-1] += [ctx.get_array(s, ARR_FACTOR) for s in (USER, ITEM)]
            if with_comm:
                prof_sum = comm.all_reduce_host([1.0, 2.5])            # host-value collective over RCCL
                assert np.array_equal(prof_sum, [1.0, 2.5])
                assert np.array_equal(comm.all_reduce_host([3.0], op="max"), [3.0])
                full = ctx.gather_user_rows(ARR_FACTOR, [0, U])        # ncclBroadcast path, one rank
                assert np.array_equal(full, ctx.get_array(USER, ARR_FACTOR))
                with pytest.raises(pmf_hip.PmfError):
                    ctx.gamma_ext_sweep(ITEM, 0.3, 0.3)
            ctx.close()
        for other in out[:2]:
            for a, b in zip(other, out[2]):
                assert np.array_equal(a, b)
    finally:
        comm.close()


def test_comm_error_paths():
    import pmf_hip
    from pmf_hip.engine import Context
    a, b = pmf_hip.Context(10, 10, 4), pmf_hip.Context(10, 10, 4)
    try:
        with pytest.raises(pmf_hip.PmfError, match="no communicator"):
            b.comm_attach(a)
        with pytest.raises(pmf_hip.PmfError, match="no communicator"):
            a.comm_allreduce_host([1.0])
        with pytest.raises(ValueError):
            a.comm_init(1, 0, b"short")
        with pytest.raises(pmf_hip.PmfError, match="rank"):
            a.comm_init(2, 5, Context.comm_unique_id())
        assert a.comm_info() == (1, 0, -1)
        a.comm_destroy()   # nothing attached: a no-op
        a.comm_barrier()   # without a communicator: a device sync
    finally:
        a.close(); b.close()


def _watchdog_worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PMF_COMM_TRANSPORT="hostshm", PMF_COMM_TIMEOUT_S="3")
    import time
    import pmf_hip
    from pmf_hip import dist as pdist
    comm = pdist.init_from_env(device=0)            # both ranks arrive here (the init ends in a barrier)
    if rank == 1:
        os._exit(0)                                  # this rank dies before its next collective
    t0 = time.time()
    try:
        comm.barrier()
        verdict = "returned"
    except pmf_hip.PmfError as e:
        verdict = f"{time.time() - t0:.1f} {e}"
    try:                                             # and the communicator stays failed instead of hanging later
        comm.all_reduce_host([1.0])
        verdict += " | second call returned"
    except pmf_hip.PmfError as e:
        verdict += f" | {e}"
    with open(os.path.join(out_dir, "watchdog.txt"), "w") as fh:
        fh.write(verdict)
    os._exit(0)


def test_a_dead_peer_ends_in_an_error_not_a_hang(tmp_path):
    """PMF_COMM_TIMEOUT_S: a rank whose peer died before the next collective gets PMF_ECOMM from the wait
    (stream + communicator watchdog) within the deadline, and every later collective call fails at once."""
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_watchdog_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
        p.join()
    assert not hung, "the surviving rank hung"
    verdict = open(os.path.join(tmp_path, "watchdog.txt")).read()
    seconds = float(verdict.split()[0])
    assert 2.0 <= seconds <= 30.0, verdict
    assert "pmf_comm_barrier failed (-5)" in verdict and ("no progress" in verdict or "peer failed" in verdict), verdict
    assert "has failed earlier" in verdict.split("|")[1] or "peer failed" in verdict.split("|")[1], verdict
