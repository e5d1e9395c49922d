"""User-sharded multi-process `fit` of the model classes: two ranks (gloo, both on
the one GPU of the test box) must reproduce the single-process fit -- factors,
validation trajectory, stop iteration -- and leave predict() working on every rank."""
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest

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


def _build(kind, comm=None):
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


def _fit(kind, model, train, val):
    if kind in ("gauss", "sgd"):
        gm = float(train["rating"].mean())
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


def _worker(rank, world, port, kind, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ["PMF_DIST_CHUNKS"] = "3"   # the messages here are too small for the default to pipeline
    import torch.distributed as tdist
    from pmf_hip import dist as pdist
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    train, val = _data()
    model, keys = _build(kind, comm=pdist.Comm())
    pred = _fit(kind, model, train, val)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), pred=pred, val_rmse=np.array(model.history_["val_rmse"]),
             iters=model.history_["iterations"], **{k: getattr(model, k) for k in keys})
    if kind == "gauss":
        np.save(os.path.join(out_dir, f"V{rank}.npy"), model.V_theta[::97])
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hpf", "poisson", "gauss"])
def test_two_rank_fit_matches_single_process(kind, tmp_path):
    import torch.multiprocessing as mp
    train, val = _data()
    model, keys = _build(kind)
    pred = _fit(kind, model, train, val)
    mp.spawn(_worker, args=(2, _free_port(), kind, str(tmp_path)), nprocs=2, join=True)
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


def test_two_rank_gradient_mode_fit_is_consistent(tmp_path):
    """The gradient mode averages the displacements of an item's pieces, so a sharded fit is a
    different (equally valid) trajectory than the single-process one: the ranks must agree with
    each other exactly and follow the same validation curve within 3 %."""
    import torch.multiprocessing as mp
    train, val = _data()
    model, keys = _build("sgd")
    _fit("sgd", model, train, val)
    mp.spawn(_worker, args=(2, _free_port(), "sgd", str(tmp_path)), nprocs=2, join=True)
    d0, d1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    for k in keys + ("pred", "val_rmse"):
        assert np.array_equal(d0[k], d1[k]), k
    assert int(d0["iters"]) == model.history_["iterations"] == 4
    np.testing.assert_allclose(d0["val_rmse"], model.history_["val_rmse"], rtol=3e-2)


@pytest.mark.parametrize("workload,chunks", [("gaussian_mf", 4), ("hpf_cavi", 4), ("gaussian_mf", 1), ("gaussian_mf_sgd", 4)])
def test_bench_two_rank_rehearsal_keeps_item_replicas_identical(workload, chunks):
    """bench.py launched as the driver launches it (torch.distributed.run, 2 ranks; gloo and one
    shared GPU stand in for RCCL over two): the replicated item state must end bit-identical on
    both ranks, i.e. kernels and collectives are ordered on the shared stream."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--small",
           "--backend", "gloo", "--share-gpu", "--only", "--workload", workload, "--chunks", str(chunks)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    assert res["config"]["item_replicas_identical"] is True


def test_pipelined_item_sweep_over_rccl_single_rank():
    """The one-GPU box cannot hold two RCCL ranks, but it can run the real backend with one:
    the pipelined item half-sweep (asynchronous all-reduce of statistic slices on RCCL's stream,
    stream-ordered waits) must give exactly what the same path gives without a collective."""
    import torch
    import torch.distributed as tdist
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_COV, ARR_FACTOR, ITEM, USER, dist as pdist
    from pmf_hip.synth import synth_ratings

    class Identity:           # same call sequence, no collective
        world = 2

        def all_reduce(self, t):
            return t

        def all_reduce_async(self, t):
            class Done:
                def wait(self):
                    pass
            return Done()

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                             device_id=dev)
    try:
        comm = pdist.Comm()
        comm.world = 2        # take the multi-rank code path; the group itself has one rank
        U, I, N, K = 20000, 3000, 600000, 64
        u, i, r = synth_ratings(U, I, N, seed=4)
        rng = np.random.default_rng(0)
        m_u, m_i = 0.1 * rng.standard_normal((U, K)), 0.1 * rng.standard_normal((I, K))
        out = []
        for c_obj in (comm, Identity()):
            ctx = pmf_hip.Context(U, I, K)
            scope = pdist.StreamScope(ctx, dev).enter()
            ctx.set_row_chunks(ITEM, 4)
            ctx.set_ratings(u, i, r - r.mean())
            ctx.set_array(USER, ARR_FACTOR, m_u); ctx.set_array(ITEM, ARR_FACTOR, m_i)
            ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
            ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))
            s_item, s_bias = pdist.gauss_stats(ctx, dev)
            for _ in range(3):
                pdist.gaussian_iteration(ctx, c_obj, s_item, s_bias, 0.5, 1.0, 1.0, 1.0)
            out.append([ctx.get_array(s, a) for s in (USER, ITEM) for a in (ARR_FACTOR, ARR_COV, ARR_BIAS)])
            g_item = pdist.gamma_stats(ctx, dev)
            ctx.set_array(USER, ARR_FACTOR, np.abs(m_u) + 0.1); ctx.set_array(ITEM, ARR_FACTOR, np.abs(m_i) + 0.1)
            ctx.set_ratings(u, i, r + 1.0)
            for _ in range(3):
                pdist.gamma_iteration(ctx, c_obj, g_item, (0.3, 0.3), (0.3, 0.3))
            out[-1] += [ctx.get_array(s, ARR_FACTOR) for s in (USER, ITEM)]
            scope.exit()
            ctx.close()
        for a, b in zip(*out):
            assert np.array_equal(a, b)
    finally:
        tdist.destroy_process_group()
