#!/usr/bin/env python3
"""Benchmark of the hot path: ratings/s per CAVI epoch on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload gaussian_mf|hpf_cavi|...]

A "step" is one full CAVI iteration (all half-sweeps of the model) over the
synthetic rating matrix of SURVEY.md section 8(d).  The default workload is
BASELINE.json configs[1]: Gaussian MF (mean-field CAVI with user/item biases,
the reference's `gaussian_mf_cavi_bias.GaussianMFCAVI`), K = 64, 1M users x
100k items, 50M ratings.

N > 1 (one process per GPU, started by `python -m torch.distributed.run` or any
launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT) STRONG-scales that
one matrix, as the north star asks and as the reference times it (one dataset, one
`fit`: compare_models.py:87-92): the ratings are sharded by user range, the item
block is replicated and the per-item sufficient statistics are all-reduced over
RCCL once per item half-sweep -- inside libpmf_hip.so (pmf_comm_init).  `--scaling
weak` gives every rank its own 1M-user / 50M-rating shard instead.  This program
never imports torch (`config.torch_imported`).

Rank 0 prints ONE JSON line (see the driver contract) with two extra objects:
`roofline` (dominant kernel: algorithmic bytes / live hipEvent time vs its bound)
and `cpu_baseline` (the CPU oracle timed on a bounded sample).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "prob-matrix-factorization_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

WORKLOADS = {
    # name: (n_users, n_items, nnz), K, hyper-parameters (best_hyperparams.txt:3,5)
    "gaussian_mf": dict(U=1_000_000, I=100_000, N=50_000_000, K=64,
                        hp=dict(sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0),
                        label="gaussian_mf_cavi_bias K=64, 1Mx100k synthetic, 50M ratings"),
    # BASELINE configs[3] per-GPU shard (10M x 1M, 500M ratings over 8 GPUs), K = 128
    "gaussian_mf_k128": dict(U=1_250_000, I=1_000_000, N=62_500_000, K=128,
                             hp=dict(sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0),
                             label="gaussian_mf_cavi_bias K=128, 1.25Mx1M synthetic, 62.5M ratings"),
    "hpf_cavi": dict(U=1_000_000, I=100_000, N=50_000_000, K=64,
                     hp=dict(a=0.3, a_prime=5.0, b_prime=5.0, c=0.3, c_prime=5.0, d_prime=5.0),
                     label="hpf_cavi K=64, 1Mx100k synthetic, 50M ratings"),
    # MAP / gradient mode of the Gaussian model (no reference counterpart, parity unpinned)
    "gaussian_mf_sgd": dict(U=1_000_000, I=100_000, N=50_000_000, K=64,
                            hp=dict(lr=0.01, sigma2=0.3, eta_theta2=0.5, eta_beta2=0.5, eta_bias2=1.0),
                            label="gaussian_mf MAP by gradient steps K=64, 1Mx100k synthetic, 50M ratings"),
}
SMALL = dict(U=100_000, I=10_000, N=5_000_000)  # --small: quick functional run
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD (= the fp32 vector peak)


def topk_measure(args, local_rank, steps=None):
    """`--workload topk`: top-10 items of 100k for a batch of query users at K = 64 from
    the dense reconstruction Theta . Beta^T -- the one MFMA-bound kernel of the product.  A step = one
    pmf_topk_items call over the batch (ids in, ranked lists out: the host legs are inside the call, the roofline
    uses the kernel's own hipEvent time)."""
    import pmf_hip
    from pmf_hip import ARR_FACTOR, ITEM, USER
    U, I, K, k = (100_000, 10_000, 64, 10) if args.small else (1_000_000, 100_000, 64, 10)
    if args.factors:
        K = args.factors
    Q = 32_768 if args.small else 262_144
    rng = np.random.default_rng(0)
    ctx = pmf_hip.Context(U, I, K, dtype="f32", device=local_rank)
    ctx.set_array(USER, ARR_FACTOR, rng.gamma(0.5, 1.0, (U, K)))
    ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(0.5, 1.0, (I, K)))
    users = rng.permutation(U)[:Q].astype(np.int32)
    for _ in range(max(args.warmup, 1)):
        ctx.topk_items(users, k)          # (the first full batch also sizes the scratch and output buffers)
    ctx.prof_enable(True)
    ctx.prof_reset()
    t0 = time.perf_counter()
    steps = steps or args.steps
    for _ in range(steps):
        items, scores = ctx.topk_items(users, k)
    elapsed = time.perf_counter() - t0
    ms, launches = ctx.prof_get()["topk"]
    flops = 2.0 * Q * I * K * steps
    ach = flops / (ms * 1e-3) / 1e12
    # spot check against a NumPy ranking (exact fp32 tables, fp64 products)
    A, B = ctx.get_array(USER, ARR_FACTOR)[users[:64]], ctx.get_array(ITEM, ARR_FACTOR)
    want = np.argsort(-(A @ B.T), axis=1, kind="stable")[:, :k]
    agree = float(np.mean(items[:64] == want))
    ctx.close()
    return {
        "metric": f"query users/sec, top-{k} of {I} items from Theta.Beta^T, K={K}", "value": Q * steps / elapsed,
        "unit": "users/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"top-{k} items for {Q} query users, {U}x{I} factors, K={K}" + (" [--small]" if args.small else ""),
                   "kernel_users_per_s": Q * steps / (ms * 1e-3), "agreement_with_numpy_ranking": agree,
                   "torch_imported": "torch" in sys.modules},
        "roofline": {"kernel": "topk_fused", "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": ms / max(launches, 1),
                     "launches": launches}}


def topk_bench(args, local_rank):
    print(json.dumps(topk_measure(args, local_rank)), flush=True)


def algorithmic_bytes(workload, U, I, N, K, elem=4):
    """bytes one epoch must move at minimum.
    Returns (total per iteration, per dominant-kernel launch summed over the two
    sides).  fp32 values, int32 indices."""
    kp = K * (K + 1) // 2
    if workload == "gaussian_mf_sgd":
        per_rating = elem * K + 12                      # gathered row + idx + rating + bias
        total = N * 2 * per_rating + (U + I) * (4 * elem * K + 16)   # row read, statistics written + read, row written
        return total, total
    if workload.startswith("gaussian_mf"):
        per_rating_factor = elem * K + elem * kp + 12   # mean row + packed cov + idx + rating + bias
        per_rating_bias = elem * K + 12
        per_row = elem * kp + elem * K + 8
        total = N * (2 * per_rating_factor + 2 * per_rating_bias) + (U + I) * per_row
        dominant = 2 * N * per_rating_factor + (U + I) * (elem * kp + elem * K)  # two accumulate launches
        return total, dominant
    per_rating = elem * K + 8
    total = N * 2 * per_rating + (U + I) * 4 * elem * K
    return total, total


def host_cpu():
    """(model string, logical cores) of the box the CPU baseline runs on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([int(p.get("num_threads", 1)) for p in threadpool_info()] or [1])
    except Exception:  # pragma: no cover
        return 1


def cpu_baseline_c1(device=0):
    """BASELINE configs[0] IN FULL -- the reference's own CPU-runnable case (10k x 2k, 200k ratings, K = 16): one
    iteration of the oracle's per-row loop for the three CAVI models on the host, and the engine's time per iteration
    on the same ratings (reference on the 8-core build container: 0.91 / 0.29 / 0.30 s per iteration)."""
    from oracle import cavi_oracle as orc
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER
    from pmf_hip.synth import BASE_SEED, synth_ratings
    U, I, N, K = 10_000, 2_000, 200_000, 16
    u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
    u64, i64 = u.astype(np.int64), i.astype(np.int64)
    idx = (orc.group_positions(u64, U), orc.group_positions(i64, I))
    out = {"sample": f"configs[0] in full: {U}x{I}, {N} ratings, K={K}, one iteration each"}
    for name in ("gaussian_bias", "poisson", "hpf"):
        if name == "gaussian_bias":
            x = r - float(r.mean())
            st = orc.init_gaussian(U, I, K, 0, True)
        elif name == "poisson":
            x = r
            st = orc.init_poisson(U, I, K, 0.1, 0.5, 0)
        else:
            x = r + 1.0
            st = orc.init_hpf(U, I, K, 0.3, 5.0, 5.0, 0.3, 5.0, 5.0, 0)
        with pmf_hip.Context(U, I, K, dtype="f32", device=device) as ctx:
            ctx.set_ratings(u, i, x)
            if name == "gaussian_bias":
                ctx.set_array(USER, ARR_FACTOR, st["m_theta"]); ctx.set_array(ITEM, ARR_FACTOR, st["m_beta"])
                ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
                ctx.set_array(USER, ARR_BIAS, st["m_user_bias"]); ctx.set_array(ITEM, ARR_BIAS, st["m_item_bias"])

                def step():
                    ctx.gauss_factor_sweep(USER, 0.3, 0.5); ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
                    ctx.gauss_bias_sweep(USER, 0.3, 1.0); ctx.gauss_bias_sweep(ITEM, 0.3, 1.0)
            else:
                ctx.set_array(USER, ARR_FACTOR, st["E_theta"]); ctx.set_array(ITEM, ARR_FACTOR, st["E_beta"])
                hier = name == "hpf"
                if hier:
                    ctx.set_array(USER, ARR_PRIOR_RATE, st["E_xi"]); ctx.set_array(ITEM, ARR_PRIOR_RATE, st["E_eta"])
                pu = (0.3, 0.0, True, st["gamma_a_xi"], 5.0) if hier else (0.1, 0.5)
                pi = (0.3, 0.0, True, st["gamma_a_eta"], 5.0) if hier else (0.1, 0.5)

                def step():
                    ctx.gamma_sweep(USER, *pu); ctx.gamma_sweep(ITEM, *pi)
            for _ in range(5):
                step()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(100):
                step()
            ctx.sync()
            gpu_s = (time.perf_counter() - t0) / 100
        t0 = time.perf_counter()
        if name == "gaussian_bias":
            orc.gaussian_iteration(st, idx, u64, i64, x, 0.3, 0.5, 0.5, 1.0)
        elif name == "poisson":
            orc.poisson_iteration(st, idx, u64, i64, x, 0.1, 0.5)
        else:
            orc.hpf_iteration(st, idx, u64, i64, x, 0.3, 5.0, 0.3, 5.0)
        cpu_s = time.perf_counter() - t0
        out[name] = {"cpu_ratings_per_s": N / cpu_s, "cpu_s_per_iteration": cpu_s, "gpu_us_per_iteration": gpu_s * 1e6,
                     "gpu_ratings_per_s": N / gpu_s}
    return out


def cpu_baseline(workload, K, hp, device=0, full=False):
    """The CPU oracle's per-row loop (the reference's loop structure) on a bounded sample of
    the same generator: ~10-30 s of CPU work with NumPy's default BLAS threading (`cores` = the
    threads its pool holds; the loop itself is interpreter-bound, one row at a time, as in the
    reference).  The engine then runs the same iteration on the same sample, so the line also
    carries the second half of BASELINE.json's metric ("val RMSE vs CPU ref")."""
    from oracle import cavi_oracle as orc
    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER
    from pmf_hip.synth import synth_ratings, train_val_split
    gauss = workload.startswith("gaussian_mf")
    if gauss:
        U, I, N = (20_000, 2_000, 1_100_000) if K <= 64 else (3_000, 300, 110_000)
    elif full:
        U, I, N = 1_000_000, 100_000, 50_000_000    # the C3 matrix itself
    else:
        U, I, N = 300_000, 30_000, 16_500_000       # the default sample: a third of C3's rows at C3's ratings per row
    u, i, r = synth_ratings(U, I, N, seed=7)
    u[0], i[0] = U - 1, I - 1
    (u, i, r), (vu, vi, vr) = train_val_split(u, i, r)
    u[0], i[0] = U - 1, I - 1
    N = len(u)
    u, i = u.astype(np.int64), i.astype(np.int64)
    idx = (orc.group_positions(u, U), orc.group_positions(i, I))
    gm = float(r.mean())
    x = r - gm if gauss else r + 1.0
    vy = vr - gm if gauss else vr + 1.0
    if gauss:
        st = orc.init_gaussian(U, I, K, 0, True)
    else:
        st = orc.init_hpf(U, I, K, hp["a"], hp["a_prime"], hp["b_prime"], hp["c"], hp["c_prime"], hp["d_prime"], 0)

    # the engine first (it needs the initial state), one iteration in fp32
    with pmf_hip.Context(U, I, K, dtype="f32", device=device) as ctx:
        ctx.set_ratings(u, i, x)
        if gauss:
            ctx.set_array(USER, ARR_FACTOR, st["m_theta"]); ctx.set_array(ITEM, ARR_FACTOR, st["m_beta"])
            ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
            ctx.set_array(USER, ARR_BIAS, st["m_user_bias"]); ctx.set_array(ITEM, ARR_BIAS, st["m_item_bias"])
            ctx.gauss_factor_sweep(USER, hp["sigma2"], hp["eta_theta2"]); ctx.gauss_factor_sweep(ITEM, hp["sigma2"], hp["eta_beta2"])
            ctx.gauss_bias_sweep(USER, hp["sigma2"], hp["eta_bias2"]); ctx.gauss_bias_sweep(ITEM, hp["sigma2"], hp["eta_bias2"])
            pred_gpu = ctx.predict(vu, vi, use_bias=True)
        else:
            ctx.set_array(USER, ARR_FACTOR, st["E_theta"]); ctx.set_array(ITEM, ARR_FACTOR, st["E_beta"])
            ctx.set_array(USER, ARR_PRIOR_RATE, st["E_xi"]); ctx.set_array(ITEM, ARR_PRIOR_RATE, st["E_eta"])
            ctx.gamma_sweep(USER, hp["a"], 0.0, True, st["gamma_a_xi"], hp["b_prime"])
            ctx.gamma_sweep(ITEM, hp["c"], 0.0, True, st["gamma_a_eta"], hp["d_prime"])
            pred_gpu = ctx.predict(vu, vi)

    t0 = time.perf_counter()
    if gauss:
        orc.gaussian_iteration(st, idx, u, i, x, hp["sigma2"], hp["eta_theta2"], hp["eta_beta2"], hp["eta_bias2"])
    else:
        orc.hpf_iteration(st, idx, u, i, x, hp["a"], hp["b_prime"], hp["c"], hp["d_prime"])
    dt = time.perf_counter() - t0
    if gauss:
        pred_cpu = orc.predict_dot(st["m_theta"], st["m_beta"], vu, vi, st["m_user_bias"], st["m_item_bias"])
    else:
        pred_cpu = orc.predict_dot(st["E_theta"], st["E_beta"], vu, vi)
    rm_cpu, rm_gpu = orc.rmse(vy, pred_cpu), orc.rmse(vy, pred_gpu)
    model, logical = host_cpu()
    scale_note = ("" if gauss else " -- the FULL C3 matrix (--cpu-full)" if full else
                  " -- a bounded sample with C3's ratings per row; the per-row loop's rate does not depend on the row "
                  "count, so this rate is also the extrapolated C3 rate (measured in full with --cpu-full)")
    out = {"value": N / dt, "unit": "ratings/s", "cores": blas_threads(), "kind": "port",
           "sample": f"1 iteration of the oracle's per-row NumPy loop, {U}x{I}, {N} ratings, K={K} "
                     f"(same generator), {dt:.1f} s, default BLAS threading" + scale_note,
           "cpu_model": model, "logical_cores": logical,
           "val_rmse_cpu": rm_cpu, "val_rmse_gpu_f32": rm_gpu, "val_rmse_abs_diff": abs(rm_cpu - rm_gpu)}
    if gauss and K == 64:
        out["c1_full"] = cpu_baseline_c1(device)
    if not gauss:
        # "best-effort CPU" (BASELINE.md section 3): the same iteration as whole-array NumPy
        # (gather + segment sums, no per-row interpreter loop), a quarter of the sample
        keep = u < U // 4
        u4, i4, x4 = u[keep], i[keep], x[keep]
        st4 = orc.init_hpf(U // 4, I, K, hp["a"], hp["a_prime"], hp["b_prime"], hp["c"], hp["c_prime"], hp["d_prime"], 0)
        idx4 = (orc.group_positions(u4, U // 4), orc.group_positions(i4, I))
        t0 = time.perf_counter()
        orc.hpf_iteration(st4, idx4, u4, i4, x4, hp["a"], hp["b_prime"], hp["c"], hp["d_prime"], orc.gamma_half_sweep_segsum)
        dv = time.perf_counter() - t0
        out["vectorised"] = {"value": len(u4) / dv, "unit": "ratings/s",
                             "sample": f"segment-sum NumPy restatement, {U // 4}x{I}, {len(u4)} ratings, {dv:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS) + ["topk"], default="gaussian_mf")
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = shard the one fixed matrix by user range (default, the north star's "
                         "problem); weak = every rank its own full-size shard")
    ap.add_argument("--small", action="store_true", help="1/10 size functional run (not a valid bench)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="skip the secondary measurements (HPF-CAVI, f64) of the default run")
    ap.add_argument("--transport", choices=["rccl", "hostshm"], default=os.environ.get("PMF_COMM_TRANSPORT", "rccl"),
                    help="rccl = one rank per GPU over xGMI; hostshm = rehearsal transport for ranks sharing one GPU")
    ap.add_argument("--chunks", type=int, default=None, help="item row chunks of the pipelined item half-sweep at "
                    "N > 1 (default PMF_DIST_CHUNKS or by message size; 1 = accumulate, then all-reduce, then finalize)")
    ap.add_argument("--exchange", choices=["auto", "allreduce", "scatter_gather"], default="auto",
                    help="N > 1: how an item half-sweep's statistics travel -- all-reduce + every rank finalises every item, or "
                         "reduce-scatter -> finalise 1/N of the items -> all-gather (auto: the latter for the Gaussian factor sweep)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: put every rank on GPU 0 (needs --transport hostshm)")
    ap.add_argument("--factors", type=int, default=None, help="exploration: override the workload's K")
    ap.add_argument("--cpu-full", action="store_true",
                    help="time the HPF cpu_baseline on the FULL C3 matrix (1M x 100k, 50M ratings, one iteration of the oracle's "
                         "per-row loop: about a minute of CPU) instead of the default 300k x 30k x 16.5M sample")
    args = ap.parse_args()

    import pmf_hip
    from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER, dist as pdist
    from pmf_hip.synth import BASE_SEED, synth_ratings

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)")
    if args.share_gpu:
        local_rank = 0
    if args.workload == "topk":
        if world != 1:
            raise SystemExit("--workload topk is a single-GPU measurement (queries are independent: replicas only)")
        return topk_bench(args, local_rank)
    if args.transport == "hostshm":
        os.environ["PMF_COMM_TRANSPORT"] = "hostshm"     # the rehearsal transport lives in the test build of the library
    comm = pdist.init_from_env(device=local_rank, transport=args.transport, exchange=args.exchange) if world > 1 else None
    strong = args.scaling == "strong"

    w = dict(WORKLOADS[args.workload])
    if args.small:
        w.update(SMALL)
    if args.factors:
        w["K"] = args.factors
        w["label"] += f" [K overridden to {args.factors}]"
    U, I, N, K, hp = w["U"], w["I"], w["N"], w["K"], w["hp"]

    # ---- data (untimed) ----------------------------------------------------
    t0 = time.time()
    lo = 0
    if comm is not None and strong:
        # ONE matrix: every rank draws the same ratings and keeps the ones of its user range
        u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
        bounds = pdist.shard_bounds(u, U, world)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        u, i, r = pdist.take_shard(u, i, r, bounds, rank)
        U_loc = hi - lo
    else:
        # weak scaling: every rank draws its own users / ratings over the same item catalogue
        u, i, r = synth_ratings(U, I, N, seed=BASE_SEED + rank, item_seed=(BASE_SEED if world > 1 else None))
        U_loc = U
    N_loc = len(u)
    t_gen = time.time() - t0
    if comm is not None:
        n_all = comm.all_reduce_host([float(N_loc), float(U_loc)])
        ratings_total, users_total = int(n_all[0]), int(n_all[1])
        centre = float(comm.all_reduce_host([float(r.sum())])[0]) / ratings_total   # the global train mean
    else:
        ratings_total, users_total, centre = N_loc, U_loc, float(r.mean())

    def run(workload, hp, steps, warmup, dtype):
        """Build the device state of `workload` on the shared ratings, run warmup + timed steps
        (device sync + barrier on both sides, max over ranks) and return the measurements."""
        elem = 4 if dtype == "f32" else 8
        sgd = workload == "gaussian_mf_sgd"
        gauss = workload.startswith("gaussian_mf") and not sgd
        ctx = pmf_hip.Context(U_loc, I, K, dtype=dtype, device=local_rank)
        if comm is not None:
            comm.attach(ctx)     # ITEM half-sweeps: accumulate -> RCCL all-reduce -> finalize inside the library
            n_chunks = args.chunks or pdist.default_item_chunks(world, pdist.item_message_bytes(ctx, gauss))
            ctx.set_row_chunks(ITEM, n_chunks)

        def fence():
            ctx.sync()
            if comm is not None:
                comm.barrier()

        rng = np.random.default_rng(42)
        t0 = time.time()
        if gauss or sgd:
            ctx.set_ratings(u, i, r - centre)           # centred as compare_models.py:54-65
        else:
            ctx.set_ratings(u, i, r + 1.0)              # +1 shift as compare_models.py:180-185
        t_csr = time.time() - t0
        mine = slice(lo, lo + U_loc) if (comm is not None and strong) else slice(0, U_loc)
        draw_u = U if (comm is not None and strong) else U_loc
        if gauss or sgd:
            # gaussian_mf_cavi_bias.py:52-67 initial state (item draws from their own stream: replicas agree)
            init_u = (0.1 * rng.standard_normal((draw_u, K)))[mine]
            init_i = 0.1 * np.random.default_rng(43).standard_normal((I, K))
            t0 = time.time()
            ctx.set_array(USER, ARR_FACTOR, init_u)
            ctx.set_array(ITEM, ARR_FACTOR, init_i)
            if gauss:
                ctx.set_cov_identity(USER, 1.0)
                ctx.set_cov_identity(ITEM, 1.0)
            ctx.set_array(USER, ARR_BIAS, np.zeros(U_loc))
            ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))
            if gauss:
                def step():
                    pdist.gaussian_iteration(ctx, comm, None, None, hp["sigma2"], hp["eta_theta2"], hp["eta_beta2"],
                                             hp["eta_bias2"])
                dominant = "gauss_accum"
            else:
                def step():
                    pdist.gaussian_sgd_iteration(ctx, comm, None, hp["lr"], hp["sigma2"], hp["eta_theta2"],
                                                 hp["eta_beta2"], hp["eta_bias2"])
                dominant = "gauss_sgd"
        else:
            # hpf_cavi.py:66-89 initial state
            init_u = ((hp["a"] + rng.gamma(1.0, 0.1, (draw_u, K))) / (hp["b_prime"] + rng.gamma(1.0, 0.1, (draw_u, K))))[mine]
            r2 = np.random.default_rng(43)
            init_i = (hp["c"] + r2.gamma(1.0, 0.1, (I, K))) / (hp["d_prime"] + r2.gamma(1.0, 0.1, (I, K)))
            t0 = time.time()
            ctx.set_array(USER, ARR_FACTOR, init_u)
            ctx.set_array(ITEM, ARR_FACTOR, init_i)
            ctx.set_array(USER, ARR_PRIOR_RATE, np.full(U_loc, (hp["a_prime"] + K * hp["a"]) / hp["b_prime"]))
            ctx.set_array(ITEM, ARR_PRIOR_RATE, np.full(I, (hp["c_prime"] + K * hp["c"]) / hp["d_prime"]))
            up = (hp["a"], 0.0, True, hp["a_prime"] + K * hp["a"], hp["b_prime"])
            ip = (hp["c"], 0.0, True, hp["c_prime"] + K * hp["c"], hp["d_prime"])

            def step():
                pdist.gamma_iteration(ctx, comm, None, up, ip)
            dominant = "gamma_sweep"

        ctx.sync()
        t_state = time.time() - t0
        for _ in range(warmup):
            step()
        fence()
        ctx.prof_enable(True)
        ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        prof = ctx.prof_get()
        ctx.prof_enable(False)
        if comm is not None:
            elapsed = float(comm.all_reduce_host([elapsed], op="max")[0])
        t0 = time.time()
        ctx.get_array(USER, ARR_FACTOR)
        item_factors = ctx.get_array(ITEM, ARR_FACTOR)
        t_pull = time.time() - t0
        consistent = None
        if comm is not None:
            # outside the timed region: the replicated item state must be bit-identical on all ranks
            digest = np.frombuffer(hashlib.sha256(item_factors.tobytes()).digest()[:8], dtype=np.uint8).astype(np.float64)
            top, bottom = comm.all_reduce_host(digest, op="max"), -comm.all_reduce_host(-digest, op="max")
            consistent = bool(np.array_equal(top, bottom))
            if not consistent and rank == 0:
                # reported, not raised: the line still carries the diagnostics, flagged invalid below
                print(f"bench.py: {workload}: the replicated item state differs between ranks -- the item half-sweep "
                      "and its collective are not ordered; the rate is not valid", file=sys.stderr, flush=True)
        del item_factors
        # roofline of the dominant kernel: this rank's launches against this rank's algorithmic bytes
        total_bytes, dom_bytes = algorithmic_bytes(workload, U_loc, I, N_loc, K, elem)
        whole_bytes, _ = algorithmic_bytes(workload, users_total if strong or comm is None else U_loc * world, I,
                                           ratings_total, K, elem)
        dom_ms, dom_n = prof[dominant]
        achieved = dom_bytes / (dom_ms / steps * 1e-3) / 1e9 if dom_n else 0.0  # bytes per epoch / kernel time per epoch
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": dom_ms / max(dom_n, 1),
                    "launches": dom_n}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not args.small and dtype == "f32" and not args.factors and comm is None:
            try:   # PMC bytes per launch from the committed rocprofv3 --pmc passes of this command, not measured in this run
                roofline["traffic"] = json.load(open(tpath)).get(f"{workload}:{dominant}")
                roofline["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes of this workload; static)"
            except Exception:
                pass
        if dominant == "gamma_sweep":
            # The gathered tables live in L2 / Infinity Cache (25.6 MB of item rows, 256 MB of user rows), so HBM
            # does not bound this kernel: its roofline is what the cache hierarchy delivers for this gather
            # pattern, measured live by the kernel's gather-only twin on the same ratings and tables.
            side_bytes = {USER: N_loc * (elem * K + 8) + U_loc * 4 * elem * K, ITEM: N_loc * (elem * K + 8) + I * 4 * elem * K}
            ceil_ms = {s: ctx.gather_ceiling_ms(s, 5) for s in (USER, ITEM)}
            peak = sum(side_bytes.values()) / (sum(ceil_ms.values()) * 1e-3) / 1e9
            # compulsory DRAM traffic: index + rating streams, row I/O, each gathered table read once
            dram = 2 * N_loc * 8 + (U_loc + I) * 4 * elem * K + (U_loc + I) * elem * K
            roofline.update({
                "bound": "cache_gather", "peak": peak, "frac": achieved / peak,
                "peak_source": "pmf_prof_gather_ceiling: the sweep kernel's memory side alone on the same ratings/tables, "
                               "live in this run (user side %.3f ms, item side %.3f ms per launch)" % (ceil_ms[USER], ceil_ms[ITEM]),
                "frac_is": "fraction of the gather-probe ceiling: the kernel against its own gather-only twin (a software probe "
                           "that shares the kernel's access shape), NOT a hardware bound -- see hw_reference",
                # hardware-derived reference beside it: the guide's measured chip-wide rates for UNIFORMLY random
                # rows of an Infinity-Cache-resident table, per side by the size of the table that side gathers from.  The
                # synthetic popularity is skewed, so the XCD L2s serve more than they would under a uniform gather: the
                # fraction can exceed 1 and says "at least as fast as the guide's uniform gather", not "above a bound"
                "hw_reference": (lambda rates: {
                    "source": "MI355X_MICROARCH.md 'Indexed rows': uniformly random rows of a 38 MB table 8.6 TB/s, of a 151 MB "
                              "table 7.4-7.9 TB/s (this workload: item table %.0f MB gathered by the user side, user table %.0f MB "
                              "by the item side)" % (I * elem * K / 1e6, U_loc * elem * K / 1e6),
                    "user_side_GBps": rates[USER], "item_side_GBps": rates[ITEM],
                    "time_at_reference_rates_over_kernel_time": sum(side_bytes[s_] / rates[s_] for s_ in (USER, ITEM)) / 1e6 / (dom_ms / steps)})(
                        {USER: 8600.0 if I * elem * K <= 64e6 else 7650.0, ITEM: 8600.0 if U_loc * elem * K <= 64e6 else 7650.0}),
                "algorithmic_GBps": achieved, "frac_of_hbm_peak_algorithmic": achieved / HBM_PEAK_GBS,
                "compulsory_dram_GBps": dram / (dom_ms / steps * 1e-3) / 1e9,
                "compulsory_dram_frac_of_hbm_peak": dram / (dom_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS})
        res = {
            "value": (ratings_total if strong or comm is None else N_loc * world) * steps / elapsed,
            "ms_per_step": elapsed / steps * 1e3, "steps": steps,
            "epoch_algorithmic_GB": whole_bytes / 1e9,
            "epoch_algorithmic_GBps": whole_bytes / (elapsed / steps) / 1e9,
            "roofline": roofline,
            "kernels_ms_per_step": {k: v[0] / steps for k, v in prof.items() if v[1]},
            "csr_build_and_upload_s": t_csr, "state_upload_s": t_state, "factor_download_s": t_pull,
            "device_GB": ctx.device_bytes() / 1e9,
            "item_replicas_identical": consistent, "item_chunks": ctx.n_chunks[ITEM],
        }
        if comm is not None:
            # compute-stream idle time waiting for a chunk's all-reduce = communication NOT hidden behind kernels
            res["comm_exposed_ms"] = prof["comm_wait"][0] / steps
            res["comm_allreduce_ms"] = prof["comm_allreduce"][0] / steps
        ctx.close()
        return res

    gauss = args.workload.startswith("gaussian_mf")
    main_res = run(args.workload, hp, args.steps, args.warmup, args.dtype)
    also = {}
    if args.workload == "gaussian_mf" and not args.only:
        # the north star names both models: HPF-CAVI on the same ratings (+1), same protocol
        also["hpf_cavi"] = run("hpf_cavi", WORKLOADS["hpf_cavi"]["hp"], max(args.steps, 10), max(args.warmup, 2), args.dtype)
        if comm is None and args.dtype == "f32":
            # the reference's own arithmetic is float64: both models in the engine's f64 (parity) mode
            also["f64"] = {"gaussian_mf": run("gaussian_mf", hp, 3, 1, "f64"),
                           "hpf_cavi": run("hpf_cavi", WORKLOADS["hpf_cavi"]["hp"], 5, 1, "f64")}
    del u, i, r
    if also and comm is None and args.dtype == "f32" and not args.factors:
        # the one MFMA-bound kernel of the product, same line as `--workload topk`
        t = topk_measure(args, local_rank, steps=3)
        also["topk"] = {key: t[key] for key in ("metric", "value", "unit", "ms_per_step", "dtype", "roofline")}
        also["topk"]["config"] = t["config"]

    if rank != 0:
        comm.barrier()
        comm.close()
        return

    def brief(res, metric, dtype):
        out = {"metric": metric, "unit": "ratings/s", "dtype": dtype, "value": res["value"], "ms_per_step": res["ms_per_step"],
               "steps": res["steps"], "epoch_algorithmic_GBps": res["epoch_algorithmic_GBps"], "roofline": res["roofline"],
               "kernels_ms_per_step": res["kernels_ms_per_step"]}
        for k in ("comm_exposed_ms", "comm_allreduce_ms"):
            if k in res:
                out[k] = res[k]
        return out

    per_gpu = " per GPU" if (world > 1 and not strong) else ""
    out = {
        "metric": (f"ratings/sec (epoch) Gaussian-MF MAP/SGD K={K}" if args.workload == "gaussian_mf_sgd" else
                   f"ratings/sec (epoch) Gaussian-MF K={K}" if gauss else f"ratings/sec (epoch) HPF-CAVI K={K}"),
        "value": main_res["value"], "unit": "ratings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": w["label"] + per_gpu + (" [--small]" if args.small else ""),
                   "n_users": users_total, "n_items": I, "ratings_total": ratings_total, "n_factors": K,
                   "ratings_on_rank0": N_loc, "users_on_rank0": U_loc,
                   "parallelism": (f"user-range rating shards x{world} of the one matrix, item statistics exchanged inside "
                                   f"libpmf_hip.so ({args.transport}), pipelined over {main_res['item_chunks']} item chunks"
                                   if world > 1 and strong else
                                   f"one full-size user shard per rank x{world}, item statistics exchanged inside "
                                   f"libpmf_hip.so ({args.transport}), pipelined over {main_res['item_chunks']} item chunks"
                                   if world > 1 else "single GPU"),
                   "epoch_algorithmic_GB": main_res["epoch_algorithmic_GB"],
                   "epoch_algorithmic_GBps": main_res["epoch_algorithmic_GBps"],
                   "torch_imported": "torch" in sys.modules},
        "roofline": main_res["roofline"],
        "kernels_ms_per_step": main_res["kernels_ms_per_step"],
        # host-buffer legs of the boundary (never part of `value`): ratings upload + index build,
        # initial state upload (float64 host arrays), factor means download
        "setup_s": {"generate": t_gen, "csr_build_and_upload": main_res["csr_build_and_upload_s"],
                    "state_upload": main_res["state_upload_s"], "factor_download": main_res["factor_download_s"]},
        "device_GB": main_res["device_GB"],
    }
    if world > 1:
        out["config"]["exchange"] = args.exchange
        out["config"]["item_replicas_identical"] = main_res["item_replicas_identical"]
        if not main_res["item_replicas_identical"]:
            out["invalid"] = "replicated item state differs between ranks"
        out["comm_exposed_ms"] = main_res["comm_exposed_ms"]
        out["comm_allreduce_ms"] = main_res["comm_allreduce_ms"]
    if also:
        out["also"] = {}
        if "hpf_cavi" in also:
            out["also"]["hpf_cavi"] = brief(also["hpf_cavi"], f"ratings/sec (epoch) HPF-CAVI K={K}", args.dtype)
        if "topk" in also:
            out["also"]["topk"] = also["topk"]
        if "f64" in also:
            out["also"]["f64"] = {"gaussian_mf": brief(also["f64"]["gaussian_mf"], f"ratings/sec (epoch) Gaussian-MF K={K}", "f64"),
                                  "hpf_cavi": brief(also["f64"]["hpf_cavi"], f"ratings/sec (epoch) HPF-CAVI K={K}", "f64")}
    if not args.no_cpu_baseline and world == 1 and args.workload != "gaussian_mf_sgd":
        out["cpu_baseline"] = cpu_baseline(args.workload, K, hp, local_rank, full=args.cpu_full)
        if "hpf_cavi" in also:
            out["also"]["hpf_cavi"]["cpu_baseline"] = cpu_baseline("hpf_cavi", K, WORKLOADS["hpf_cavi"]["hp"], local_rank,
                                                                   full=args.cpu_full)
    if also:
        # LAST key of the line: every headline number in compact form, so that a record that keeps only the head and
        # the tail of the line still carries them (`also` holds the detail)
        def short(res):
            rf = res["roofline"]
            return {"value": float("%.4g" % res["value"]), "ms_per_step": float("%.4g" % res["ms_per_step"]),
                    "roofline": {"bound": rf["bound"], "frac": float("%.3g" % rf["frac"]), "peak": float("%.4g" % rf["peak"])}}
        summary = {}
        if "hpf_cavi" in also:
            summary["hpf_cavi_" + args.dtype] = short(also["hpf_cavi"])
        if "f64" in also:
            summary["gaussian_mf_f64"] = short(also["f64"]["gaussian_mf"])
            summary["hpf_cavi_f64"] = short(also["f64"]["hpf_cavi"])
        if "topk" in also:
            summary["topk"] = short(also["topk"])
        out["summary"] = summary
    print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
