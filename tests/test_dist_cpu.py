"""Multi-process (world_size 2, gloo, CPU) tests of the user-sharded iteration in pmf_hip.dist:
shard -> local half-sweeps -> all-reduce of the item statistics -> finalise, against the unsharded
oracle.  On a GPU the collective runs inside libpmf_hip.so (pmf_comm_init); here the same iteration
functions drive an oracle-backed engine through their external-collective form (accumulate ->
collective -> finalize sequenced by the host), and `test_sharded_model_fit_*` runs the model
classes' sharded `fit` host logic (sharding, monitor all-reduce, gather) the same way."""
import os
import socket
import sys

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    from helpers import skewed_problem
    return skewed_problem(11, 400, 60, 6000, rating_kind="count")


def _worker(rank, world, port, kind, out_dir, chunks=1):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as tdist
    from oracle import cavi_oracle as orc
    from oracle_engine import CpuStats, GlooComm, OracleEngine
    from pmf_hip import dist as pdist
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    comm = GlooComm()
    u, i, x = _problem()
    U, I, K = 400, 60, 6
    bounds = pdist.shard_bounds(u, U, world)
    lu, li, lx = pdist.take_shard(u, i, x, bounds, rank)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    eng = OracleEngine(hi - lo, I, K, lu, li, lx)
    eng.set_row_chunks(1, chunks)  # item side: pipelined accumulate / all-reduce / finalize
    if kind == "hpf":
        st = orc.init_hpf(U, I, K, 0.3, 5.0, 5.0, 0.3, 5.0, 5.0, seed=1)
        eng.st = {"E_theta": st["E_theta"][lo:hi], "E_beta": st["E_beta"], "E_xi": st["E_xi"][lo:hi],
                  "E_eta": st["E_eta"]}
        stats = CpuStats(I * 2 * K)
        up = (0.3, 0.0, True, st["gamma_a_xi"], 5.0)
        ip = (0.3, 0.0, True, st["gamma_a_eta"], 5.0)
        for _ in range(3):
            pdist.gamma_iteration(eng, comm, stats, up, ip)
        res = {"E_theta": eng.st["E_theta"], "E_beta": eng.st["E_beta"], "E_xi": eng.st["E_xi"],
               "E_eta": eng.st["E_eta"]}
    elif kind == "sgd":
        eng.x = lx - 4.0
        st = orc.init_gaussian(U, I, K, seed=1, bias=True)
        eng.st = {"m_theta": st["m_theta"][lo:hi], "m_beta": st["m_beta"], "m_user_bias": st["m_user_bias"][lo:hi],
                  "m_item_bias": st["m_item_bias"]}
        s_item = CpuStats(I * eng.sgd_stats_width)
        for _ in range(3):
            pdist.gaussian_sgd_iteration(eng, comm, s_item, 0.02, 0.5, 1.0, 1.0, 1.5)
        res = {k: eng.st[k] for k in ("m_theta", "m_beta", "m_user_bias", "m_item_bias")}
    else:
        xc = lx - 4.0
        eng.x = xc
        st = orc.init_gaussian(U, I, K, seed=1, bias=True)
        eng.st = {"m_theta": st["m_theta"][lo:hi], "V_theta": st["V_theta"][lo:hi], "m_beta": st["m_beta"],
                  "V_beta": st["V_beta"], "m_user_bias": st["m_user_bias"][lo:hi], "m_item_bias": st["m_item_bias"]}
        s_item, s_bias = CpuStats(I * (K * K + K)), CpuStats(I * 2)
        for _ in range(3):
            pdist.gaussian_iteration(eng, comm, s_item, s_bias, 0.3, 0.5, 0.5, 1.0)
        res = {k: eng.st[k] for k in ("m_theta", "m_beta", "V_beta", "m_user_bias", "m_item_bias")}
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, **res)
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("chunks", [1, 3])
@pytest.mark.parametrize("kind", ["hpf", "gauss"])
def test_two_rank_sharded_iteration_matches_unsharded_oracle(kind, chunks, tmp_path):
    import torch.multiprocessing as mp
    from oracle import cavi_oracle as orc
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), chunks), nprocs=world, join=True)
    u, i, x = _problem()
    U, I, K = 400, 60, 6
    idx = (orc.group_positions(u, U), orc.group_positions(i, I))
    if kind == "hpf":
        st = orc.init_hpf(U, I, K, 0.3, 5.0, 5.0, 0.3, 5.0, 5.0, seed=1)
        for _ in range(3):
            orc.hpf_iteration(st, idx, u, i, x, 0.3, 5.0, 0.3, 5.0, orc.gamma_half_sweep_segsum)
        user_keys, item_keys = ["E_theta", "E_xi"], ["E_beta", "E_eta"]
    else:
        st = orc.init_gaussian(U, I, K, seed=1, bias=True)
        for _ in range(3):
            orc.gaussian_iteration(st, idx, u, i, x - 4.0, 0.3, 0.5, 0.5, 1.0, vectorised=True)
        user_keys, item_keys = ["m_theta", "m_user_bias"], ["m_beta", "V_beta", "m_item_bias"]
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert parts[0]["lo"] == 0 and parts[0]["hi"] == parts[1]["lo"] and parts[1]["hi"] == U
    for k in user_keys:
        got = np.concatenate([p[k] for p in parts], axis=0)
        np.testing.assert_allclose(got, st[k], rtol=1e-10, atol=1e-12, err_msg=k)
    for k in item_keys:
        for p in parts:  # replicated item state is identical on every rank
            np.testing.assert_allclose(p[k], st[k], rtol=1e-10, atol=1e-12, err_msg=k)
        assert np.array_equal(parts[0][k], parts[1][k])


def test_shard_bounds_balance_ratings_not_rows():
    sys.path.insert(0, os.path.join(ROOT, "prob-matrix-factorization_amd"))
    from pmf_hip import dist as pdist
    u, i, x = _problem()
    for world in (1, 2, 3, 8):
        b = pdist.shard_bounds(u, 400, world)
        assert b[0] == 0 and b[-1] == 400 and np.all(np.diff(b) >= 0)
        per = [np.sum((u >= b[g]) & (u < b[g + 1])) for g in range(world)]
        assert sum(per) == len(u)
        assert max(per) - min(per) <= np.bincount(u).max() + 1
        total = 0
        for g in range(world):
            lu, li, lx = pdist.take_shard(u, i, x, b, g)
            assert lu.min(initial=0) >= 0 and lu.max(initial=0) < max(b[g + 1] - b[g], 1)
            total += len(lu)
        assert total == len(u)


def test_shard_bounds_never_leave_a_rank_without_users():
    sys.path.insert(0, os.path.join(ROOT, "prob-matrix-factorization_amd"))
    from pmf_hip import dist as pdist
    # all ratings on two users at the two ends: rating balance alone would give empty ranges
    u = np.array([0] * 50 + [9] * 50)
    for world in (2, 3, 5, 10):
        b = pdist.shard_bounds(u, 10, world)
        assert b[0] == 0 and b[-1] == 10 and np.all(np.diff(b) >= 1)
    with pytest.raises(ValueError):
        pdist.shard_bounds(u, 10, 11)


@pytest.mark.parametrize("chunks", [1, 3])
def test_two_rank_sgd_epochs_follow_the_partitioned_definition(chunks, tmp_path):
    """Gradient mode over two ranks (gloo): users local, the items' rating-count-weighted
    displacement sums all-reduced -- against the same definition evaluated in one process
    (a half-sweep per shard from the old item values, displacements averaged by count)."""
    import torch.multiprocessing as mp
    from oracle import cavi_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "prob-matrix-factorization_amd"))
    from pmf_hip import dist as pdist
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), "sgd", str(tmp_path), chunks), nprocs=world, join=True)
    u, i, x = _problem()
    x = x - 4.0
    U, I, K = 400, 60, 6
    st = orc.init_gaussian(U, I, K, seed=1, bias=True)
    bounds = pdist.shard_bounds(u, U, world)
    shards = [pdist.take_shard(u, i, x, bounds, g) for g in range(world)]
    for _ in range(3):
        new_t, new_bu = st["m_theta"].copy(), st["m_user_bias"].copy()
        for g, (lu, li, lx) in enumerate(shards):            # user side: local to the shard
            lo, hi = int(bounds[g]), int(bounds[g + 1])
            ptr, pos = orc.group_positions(lu, hi - lo)
            f, b = orc.gauss_sgd_half_sweep(st["m_theta"][lo:hi], st["m_beta"], st["m_user_bias"][lo:hi],
                                            st["m_item_bias"], ptr, pos, li, lx, 0.02, 0.5, 1.0, 1.5)
            new_t[lo:hi], new_bu[lo:hi] = f, b
        st["m_theta"], st["m_user_bias"] = new_t, new_bu
        num_f, num_b, cnt = np.zeros((I, K)), np.zeros(I), np.zeros(I)
        for g, (lu, li, lx) in enumerate(shards):            # item side: weighted mean over the shards
            lo, hi = int(bounds[g]), int(bounds[g + 1])
            ptr, pos = orc.group_positions(li, I)
            f, b = orc.gauss_sgd_half_sweep(st["m_beta"], st["m_theta"][lo:hi], st["m_item_bias"],
                                            st["m_user_bias"][lo:hi], ptr, pos, lu, lx, 0.02, 0.5, 1.0, 1.5)
            n = np.diff(ptr).astype(float)
            num_f += n[:, None] * (f - st["m_beta"]); num_b += n * (b - st["m_item_bias"]); cnt += n
        live = cnt > 0
        st["m_beta"][live] += num_f[live] / cnt[live, None]
        st["m_item_bias"][live] += num_b[live] / cnt[live]
    parts = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    got_t = np.concatenate([p["m_theta"] for p in parts])
    assert np.max(np.abs(got_t - st["m_theta"])) <= 1e-12
    assert np.max(np.abs(np.concatenate([p["m_user_bias"] for p in parts]) - st["m_user_bias"])) <= 1e-12
    for p in parts:
        assert np.max(np.abs(p["m_beta"] - st["m_beta"])) <= 1e-12
        assert np.max(np.abs(p["m_item_bias"] - st["m_item_bias"])) <= 1e-12


# ---- the model classes' sharded fit, host logic only (no GPU): OracleContext stands in for pmf_hip.Context -----
def _fit_problem():
    import pandas as pd
    from helpers import skewed_problem
    u, i, x = skewed_problem(5, 300, 50, 5000, rating_kind="count")
    rng = np.random.default_rng(0)
    is_val = rng.random(len(u)) < 0.15
    is_val[0] = False
    train = pd.DataFrame({"u": u[~is_val], "i": i[~is_val], "rating": x[~is_val]})
    val = pd.DataFrame({"u": np.append(u[is_val], 305), "i": np.append(i[is_val], 3), "rating": np.append(x[is_val], 4.0)})
    return train, val


def _fit_model(kind, comm):
    if kind == "hpf":
        from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
        return HPF_CAVI(HPF_CAVI_Config(n_factors=5, a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0,
                                        max_iter=8, tol=1e-3, verbose=False), dtype="f64", comm=comm), \
            ("E_theta", "E_beta", "E_xi", "E_eta", "gamma_a_theta", "gamma_b_beta")
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    return GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=5, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0,
                                               max_iter=4, tol=-1.0, verbose=False), dtype="f64", comm=comm), \
        ("m_theta", "m_beta", "m_user_bias", "m_item_bias")


def _run_fit(kind, comm, presharded=False):
    train, val = _fit_problem()
    model, keys = _fit_model(kind, comm)
    gm = float(train["rating"].mean())
    full_val = val
    if presharded:      # hand this rank only its own rows (global user ids)
        from pmf_hip import dist as pdist
        b = pdist.shard_bounds(train["u"].to_numpy(), int(train["u"].max()) + 1, comm.world)
        lo, hi = int(b[comm.rank]), int(b[comm.rank + 1])
        train = train[(train["u"] >= lo) & (train["u"] < hi)]
        val = val[(val["u"] >= lo) & ((val["u"] < hi) | (comm.rank == comm.world - 1))]
        model = type(model)(model.config, dtype="f64", comm=comm, presharded=True)
    if kind == "gauss":
        a, b = train.copy(), val.copy()
        a["rating"] -= gm; b["rating"] -= gm
        model.fit(a, val_df=b, global_mean=gm)
        pred = model.predict(full_val["u"].to_numpy(), full_val["i"].to_numpy(), gm)
    else:
        model.fit(train, val_df=val)
        pred = model.predict(full_val["u"].to_numpy(), full_val["i"].to_numpy())
    out = {k: getattr(model, k) for k in keys}
    out.update(pred=pred, val_rmse=np.array(model.history_["val_rmse"]), iters=model.history_["iterations"])
    return model, out


def _fit_worker(rank, world, port, kind, out_dir, presharded=False):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      PMF_DIST_CHUNKS="3")
    import torch.distributed as tdist
    import pmf_hip
    from oracle_engine import AttachingGlooComm, OracleContext
    pmf_hip.Context = OracleContext           # the model classes look the context class up at run time
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    model, out = _run_fit(kind, AttachingGlooComm() if world > 1 else None, presharded and world > 1)
    if kind == "gauss" and world > 1:
        lo, hi = model.user_range
        assert model.V_theta.shape[0] == hi - lo                  # local rows only; the gather is explicit
        out["V_theta"] = model.gather_V_theta()
    np.savez(os.path.join(out_dir, f"w{world}{'p' if presharded else ''}_rank{rank}.npz"), **out)
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hpf", "gauss"])
def test_sharded_model_fit_host_logic_over_gloo(kind, tmp_path):
    """Two gloo ranks run the model classes' user-sharded `fit` (shard_bounds / take_shard, the attached
    communicator, the all-reduced validation monitor with identical early-stop decisions, the gather of the
    user side, predict on the full-size context) and must reproduce the one-rank fit."""
    import torch.multiprocessing as mp
    for world, pre in ((1, False), (2, False), (2, True)):
        mp.spawn(_fit_worker, args=(world, _free_port(), kind, str(tmp_path), pre), nprocs=world, join=True)
    one = np.load(os.path.join(tmp_path, "w1_rank0.npz"))
    for rank, tag in ((0, "w2"), (1, "w2"), (0, "w2p"), (1, "w2p")):     # full frames on every rank / presharded frames
        d = np.load(os.path.join(tmp_path, f"{tag}_rank{rank}.npz"))
        assert int(d["iters"]) == int(one["iters"]) and int(one["iters"]) >= 3
        np.testing.assert_allclose(d["val_rmse"], one["val_rmse"], rtol=1e-10)
        for k in one.files:
            if k not in ("iters", "val_rmse"):
                np.testing.assert_allclose(d[k], one[k], rtol=1e-9, atol=1e-11, err_msg=k)
    if kind == "gauss":
        assert np.load(os.path.join(tmp_path, "w2_rank0.npz"))["V_theta"].shape == (300, 5, 5)


def _contract_worker(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as tdist
    import pmf_hip
    from oracle_engine import AttachingGlooComm, OracleContext
    from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
    pmf_hip.Context = OracleContext
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    comm = AttachingGlooComm()
    rng = np.random.default_rng(5)
    # users 0, 1 | 2 (no training rows) | 3, 4, 5; explicit ranges [0, 3) and [3, 6); derived ones [0, 2) and [2, 6)
    tu = np.array([0, 0, 1, 1, 1, 3, 4, 4, 5, 5, 5, 3]); ti = rng.integers(0, 4, len(tu)); ti[0] = 3
    train = pd.DataFrame({"u": tu, "i": ti, "rating": rng.integers(0, 6, len(tu)).astype(float)})
    gm = float(train["rating"].mean())
    cfg = GaussianMFCAVIConfig(n_factors=3, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0, max_iter=3,
                               tol=-1.0, verbose=False)
    mine = train[train["u"] < 3] if rank == 0 else train[train["u"] >= 3]
    centred = lambda df: df.assign(rating=df["rating"] - gm)      # noqa: E731
    out = {}

    def sharded(val, presharded, split):
        lo, hi = split[rank], split[rank + 1]
        part = val[(val["u"] >= lo) & ((val["u"] < hi) | (rank == world - 1))]
        m = GaussianMFCAVI(cfg, dtype="f64", comm=comm, presharded=presharded)
        m.fit(centred(mine), centred(part), global_mean=gm)
        return np.array(m.history_["val_rmse"]), m.m_theta

    def single(val):
        m = GaussianMFCAVI(cfg, dtype="f64")
        m.fit(centred(train), centred(val), global_mean=gm)
        return np.array(m.history_["val_rmse"]), m.m_theta

    # A: every validation row belongs to rank 0 -- rank 1's share is empty and must still join the collectives
    val_a = pd.DataFrame({"u": [0, 1, 1, 9], "i": [0, 1, 2, 1], "rating": [4.0, 5.0, 3.0, 2.0]})
    val_a = val_a[val_a["u"] < 6]                       # (the unseen id would go to the last rank)
    got, theta = sharded(val_a, True, [0, 2, 6])
    want, theta1 = single(val_a)
    np.testing.assert_allclose(got, want, rtol=1e-10)
    np.testing.assert_allclose(theta, theta1, rtol=1e-9, atol=1e-12)
    # B: user 2 has no training rows; explicit ranges put it on rank 0, and so do the validation rows
    val_b = pd.DataFrame({"u": [2, 0, 4, 5, 7], "i": [0, 1, 2, 1, 0], "rating": [4.0, 5.0, 3.0, 2.0, 1.0]})
    got, _ = sharded(val_b, np.array([0, 3, 6]), [0, 3, 6])
    want, _ = single(val_b)
    np.testing.assert_allclose(got, want, rtol=1e-10)
    # C: the same rows under DERIVED ranges ([0, 2) | [2, 6)): user 2's row sits on the wrong rank -> refused everywhere
    try:
        sharded(val_b, True, [0, 3, 6])
        out["stray"] = "accepted"
    except ValueError as e:
        out["stray"] = str(e)
    # D: explicit ranges that do not contain the rows -> refused everywhere
    try:
        sharded(val_a, np.array([0, 1, 6]), [0, 2, 6])
        out["bad_bounds"] = "accepted"
    except ValueError as e:
        out["bad_bounds"] = str(e)
    with open(os.path.join(out_dir, f"contract{rank}.txt"), "w") as fh:
        fh.write(out["stray"] + "\n" + out["bad_bounds"] + "\n")
    tdist.barrier()
    tdist.destroy_process_group()


def test_presharded_contract_over_gloo(tmp_path):
    """Presharded fits: a rank with no validation rows keeps step with the others (its share of the all-reduced
    sums is zero); explicit user ranges (`presharded=bounds`) place users without training rows; rows on the
    wrong rank are refused on every rank instead of dropping out of the metric."""
    import torch.multiprocessing as mp
    mp.spawn(_contract_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        stray, bad = open(os.path.join(tmp_path, f"contract{rank}.txt")).read().splitlines()
        assert "1 validation row(s) were given to a rank that does not own" in stray
        assert "training row(s) lie outside their rank's user range" in bad


# ---- the communicator-id rendezvous (pmf_hip/dist.py:exchange_unique_id) -------------------------------
_RDV = r'''
import os, sys, time
sys.path[:0] = [{pkg!r}]
from pmf_hip import dist
rank, world, timeout = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
try:
    uid, path = dist.exchange_unique_id(rank, lambda: bytes([65 + rank]) * 128, timeout=timeout, world=world)
    print("OK", uid[:1].decode(), flush=True)
except TimeoutError as e:
    print("TIMEOUT", e, flush=True)
    sys.exit(3)
'''


def _rdv(rank, world, timeout, id_file):
    import subprocess
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prob-matrix-factorization_amd")
    env = dict(os.environ, PMF_COMM_ID_FILE=id_file)
    return subprocess.Popen([sys.executable, "-c", _RDV.format(pkg=pkg), str(rank), str(world), str(timeout)], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def test_id_rendezvous_ignores_a_stale_file_and_hands_every_rank_the_new_id(tmp_path):
    """A crashed launch left an id file (and acknowledgements) at the path the next launch of the same shell / port
    uses: the new ranks must end with rank 0's NEW id, never with the leftover."""
    import time
    base = str(tmp_path / "uid")
    old = time.time() - 3600
    for name, content in ((base + ".0", b"Z" * 128), (base + ".0.ack1", b"Z" * 128)):
        with open(name, "wb") as f:
            f.write(content)
        os.utime(name, (old, old))
    late = _rdv(1, 2, 20, base)          # rank 1 arrives first and finds only the stale file
    time.sleep(1.0)
    first = _rdv(0, 2, 20, base)
    out0, out1 = first.communicate(timeout=60)[0], late.communicate(timeout=60)[0]
    assert first.returncode == 0 and late.returncode == 0, (out0, out1)
    assert out0.strip() == "OK A" and out1.strip() == "OK A", (out0, out1)


def test_id_rendezvous_has_a_deadline_for_a_peer_that_never_starts(tmp_path):
    """Rank 0 of a 2-rank launch whose peer never starts: a TimeoutError before anybody calls ncclCommInitRank
    (which has no deadline of its own), not a hang; likewise a rank 1 that only ever sees a stale id."""
    import time
    base = str(tmp_path / "uid")
    alone = _rdv(0, 2, 1.5, base)
    out = alone.communicate(timeout=60)[0]
    assert alone.returncode == 3 and "never acknowledged" in out and "[1]" in out, out
    stale = str(tmp_path / "uid2")
    with open(stale + ".0", "wb") as f:
        f.write(b"Z" * 128)
    old = time.time() - 3600
    os.utime(stale + ".0", (old, old))
    orphan = _rdv(1, 2, 1.5, stale)
    out = orphan.communicate(timeout=60)[0]
    assert orphan.returncode == 3 and "no fresh communicator id" in out, out
