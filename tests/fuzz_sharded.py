"""Randomised sweep of SHARDED fits: every rank of a `hostshm` run (ranks sharing the one GPU of the test box)
draws the same random small problems as tests/fuzz_parity.py, fits them sharded by user range -- full-frame and
presharded, random item row chunks -- and rank 0 compares the result with a single-context fit of the same model.
Ranks with no users, one-item catalogues, rows longer than a task and out-of-range validation ids all occur.
Test infrastructure; launched by tests/test_fuzz_gpu.py (spawned workers, no torch in them)."""
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"hpf": ("E_theta", "E_beta", "E_xi", "E_eta", "gamma_a_theta", "gamma_b_beta"),
        "poisson": ("E_theta", "E_beta", "a_theta", "b_beta"),
        "gauss_bias": ("m_theta", "m_beta", "m_user_bias", "m_item_bias"),
        "gauss": ("m_theta", "m_beta")}


def build(kind, K, seed, iters, comm=None, presharded=False):
    kw = dict(dtype="f64", comm=comm, presharded=presharded) if comm is not None else dict(dtype="f64")
    if kind == "hpf":
        from src.models.hpf_cavi import HPF_CAVI, HPF_CAVI_Config
        return HPF_CAVI(HPF_CAVI_Config(n_factors=K, a=0.3, a_prime=2.0, b_prime=1.5, c=0.4, c_prime=3.0, d_prime=0.7,
                                        max_iter=iters, tol=None, random_state=seed, verbose=False), **kw)
    if kind == "poisson":
        from src.models.poisson_mf_cavi import PoissonMFCAVI, PoissonMFCAVIConfig
        return PoissonMFCAVI(PoissonMFCAVIConfig(n_factors=K, a0=0.2, b0=0.6, max_iter=iters, tol=None,
                                                 random_state=seed, verbose=False), **kw)
    if kind == "gauss_bias":
        from src.models.gaussian_mf_cavi_bias import GaussianMFCAVI, GaussianMFCAVIConfig
        return GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=K, sigma2=0.4, eta_theta2=0.6, eta_beta2=0.9, eta_bias2=1.3,
                                                   max_iter=iters, tol=-1e9, random_state=seed, verbose=False), **kw)
    from src.models.gaussian_mf_cavi import GaussianMFCAVI, GaussianMFCAVIConfig
    return GaussianMFCAVI(GaussianMFCAVIConfig(n_factors=K, sigma2=0.4, eta_theta2=0.6, eta_beta2=0.9, max_iter=iters,
                                               tol=-1e9, random_state=seed, verbose=False), **kw)


def fit(kind, model, train, val, gm):
    if kind.startswith("gauss"):
        model.fit(train.assign(rating=train["rating"] - gm), val.assign(rating=val["rating"] - gm), global_mean=gm)
    else:
        model.fit(train.assign(rating=train["rating"] + 1.0), val.assign(rating=val["rating"] + 1.0))


def worker(rank, world, port, n_trials, seed, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", PMF_COMM_TRANSPORT="hostshm")
    from fuzz_parity import problem
    from pmf_hip import dist as pdist
    comm = pdist.init_from_env(device=0)
    rng = np.random.default_rng(seed)          # the same stream on every rank: the same problems
    failures, worst, lines = 0, 0.0, []
    for t in range(n_trials):
        kind = str(rng.choice(list(KEYS)))
        K = int(rng.choice([1, 3, 8, 12, 20, 33, 64, 72]))
        shape, u, i, x, (vu, vi, vx) = problem(rng)
        seed_t, iters = int(rng.integers(0, 1000)), int(rng.integers(1, 4))
        pre = bool(rng.integers(0, 2))
        explicit = bool(rng.integers(0, 2))
        os.environ["PMF_DIST_CHUNKS"] = str(int(rng.integers(1, 5)))
        comm.exchange = [None, "allreduce", "scatter_gather"][int(rng.integers(0, 3))]   # applied when the model attaches its context
        train = pd.DataFrame({"u": u, "i": i, "rating": x})
        val = pd.DataFrame({"u": vu, "i": vi, "rating": vx})
        gm = float(train["rating"].mean())
        mine_t, mine_v = train, val
        tag = f"trial {t}: {kind} K={K} shape={shape} U={int(u.max()) + 1} I={int(i.max()) + 1} N={len(u)} pre={pre} explicit={explicit} " \
              f"chunks={os.environ['PMF_DIST_CHUNKS']} exchange={comm.exchange} iters={iters}"
        try:
            arg = False
            if pre:      # this rank is handed only its own rows (global ids)
                n_users = int(train["u"].max()) + 1
                b = pdist.shard_bounds(train["u"].to_numpy(), n_users, world)
                lo, hi = int(b[rank]), int(b[rank + 1])
                mine_t = train[(train["u"] >= lo) & (train["u"] < hi)]
                arg = b if explicit else True
                if not explicit:
                    # ranges derived from the rows: a user without training rows belongs to the next rank above
                    # that has some -- the validation rows must follow the same rule
                    tops = [int(train["u"][(train["u"] >= b[r]) & (train["u"] < b[r + 1])].max()) + 1
                            if ((train["u"] >= b[r]) & (train["u"] < b[r + 1])).any() else 0 for r in range(world)]
                    d = np.concatenate([[0], np.maximum.accumulate(tops)])
                    d[-1] = n_users
                    lo, hi = int(d[rank]), int(d[rank + 1])
                mine_v = val[(val["u"] >= lo) & ((val["u"] < hi) | (rank == world - 1))]
            model = build(kind, K, seed_t, iters, comm=comm, presharded=arg)
            fit(kind, model, mine_t, mine_v, gm)
            got = {k: np.asarray(getattr(model, k)) for k in KEYS[kind]}
            hist = np.asarray(model.history_["val_rmse"], dtype=np.float64)
            model.close()
            if rank == 0:
                ref = build(kind, K, seed_t, iters)
                fit(kind, ref, train, val, gm)
                err = 0.0
                for k in KEYS[kind]:
                    want = np.asarray(getattr(ref, k))
                    assert got[k].shape == want.shape, (k, got[k].shape, want.shape)
                    scale = max(1.0, float(np.max(np.abs(want))) if want.size else 1.0)
                    err = max(err, float(np.max(np.abs(got[k] - want))) / scale if want.size else 0.0)
                rh = np.asarray(ref.history_["val_rmse"], dtype=np.float64)
                assert hist.shape == rh.shape, (hist, rh)
                same = (np.isnan(hist) & np.isnan(rh)) | (np.abs(hist - rh) <= 1e-9 * np.maximum(1.0, np.abs(rh)))
                assert np.all(same), (hist, rh)
                ref.close()
                if not err <= 1e-9:
                    raise AssertionError(f"deviation {err:.3e}")
                worst = max(worst, err)
        except ValueError as e:    # fewer users than ranks / a rank without a user range: refused on every rank alike
            if "rank" not in str(e):
                raise
            lines.append(f"refused {tag}: {e}")
        except Exception as e:     # noqa: BLE001 -- reported; the ranks stay in step (every rank runs every collective or none)
            failures += 1
            lines.append(f"FAIL rank {rank} {tag}: {type(e).__name__}: {e}")
            break                  # after a failure the ranks may be out of step: stop the sweep
    assert "torch" not in sys.modules
    with open(f"{out_path}.rank{rank}", "w") as fh:
        fh.write("\n".join(lines + [f"done {t + 1} {failures} {worst:.3e}"]) + "\n")
    comm.barrier()
    comm.close()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":      # python tests/fuzz_sharded.py [world] [n_trials] [seed]: a long sweep, logs printed
    import multiprocessing as mp
    import socket
    import tempfile
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = os.path.join(tempfile.mkdtemp(), "sweep")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, world, port, n, seed, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    for p in procs:
        if p.is_alive():
            p.kill()
    for r in range(world):
        if os.path.exists(f"{out}.rank{r}"):
            text = open(f"{out}.rank{r}").read().splitlines()
            print(f"rank {r}: {sum(l.startswith('refused') for l in text)} refused;", *[l for l in text if not l.startswith("refused")])
            if r == 0:
                import collections
                why = collections.Counter(l.split(": ", 2)[-1][:60] for l in text if l.startswith("refused"))
                for reason, count in why.most_common():
                    print(f"    {count:4d} x {reason}")
    sys.exit(max((p.exitcode or 0) for p in procs))
