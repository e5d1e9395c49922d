"""Per-GPU cost of the multi-GPU code path with the REAL RCCL calls in it, on one GPU: a C2 / C3-sized context with a
one-rank RCCL communicator attached (ITEM half-sweeps = accumulate -> collective per chunk on the collective
stream -> finalize) against the same context without one (fused sweeps).
    python tools/probe_comm_single.py [gauss|hpf|gauss_k128] [chunks ...]
`gauss_k128` = BASELINE config C4's per-GPU shard (K = 128, 1.25M x 1M, 62.5M ratings): the path a C4 rank really runs.
Three forms are timed: fused (no communicator); item side in three stages (accumulate-only kernel, collective, row solves
on the finalize stream) with the user side fused; the same with the user side un-fused too (PMF_GAUSS_UNFUSED:
accumulate-only + a separate solve launch on the compute stream).  Both exchanges (all-reduce; reduce-scatter ->
finalize -> all-gather, which with one rank finalises everything too) are run."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd")]
import pmf_hip  # noqa: E402
from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER, dist as pdist  # noqa: E402
from pmf_hip.engine import Context  # noqa: E402
from pmf_hip.synth import BASE_SEED, synth_ratings  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "gauss"
    chunk_list = [int(a) for a in sys.argv[2:]] or [1, 4, 8]
    U, I, N, K = 1_000_000, 100_000, 50_000_000, 64
    if kind == "gauss_k128":
        U, I, N, K = 1_250_000, 1_000_000, 62_500_000, 128
    u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
    comm = pdist.Comm(0, 1, 0, Context.comm_unique_id(), "rccl")
    rng = np.random.default_rng(1)
    forms = [(0, "auto", False)] + [(c, ex, False) for c in chunk_list for ex in ("allreduce", "scatter_gather")]
    if kind == "gauss_k128":
        forms += [(0, "auto", True)] + [(c, "allreduce", True) for c in chunk_list]
    for chunks, exchange, unfused in forms:     # chunks 0 = no communicator (fused sweeps)
        os.environ.pop("PMF_GAUSS_UNFUSED", None)
        if unfused:
            os.environ["PMF_GAUSS_UNFUSED"] = "1"       # read when the context is created
        ctx = pmf_hip.Context(U, I, K)
        if chunks:
            comm.attach(ctx)
            ctx.comm_set_exchange(exchange)
            ctx.set_row_chunks(ITEM, chunks)
        if kind.startswith("gauss"):
            ctx.set_ratings(u, i, r - r.mean())
            ctx.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K)))
            ctx.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
            ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
            ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))

            def step():
                pdist.gaussian_iteration(ctx, comm if chunks else None, None, None, 0.3, 0.5, 0.5, 1.0)
        else:
            ctx.set_ratings(u, i, r + 1.0)
            ctx.set_array(USER, ARR_FACTOR, rng.gamma(1.0, 0.3, (U, K)) + 0.1)
            ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(1.0, 0.3, (I, K)) + 0.1)
            ctx.set_array(USER, ARR_PRIOR_RATE, np.full(U, 1.0)); ctx.set_array(ITEM, ARR_PRIOR_RATE, np.full(I, 1.0))
            up = ip = (0.3, 0.0, True, 0.3 + K * 0.3, 1.0)

            def step():
                pdist.gamma_iteration(ctx, comm if chunks else None, None, up, ip)
        reps = 3 if kind == "gauss_k128" else 5
        for _ in range(1 if kind == "gauss_k128" else 2):
            step()
        ctx.sync()
        ctx.prof_enable(True); ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        ctx.sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        prof = {k: round(v[0] / reps, 3) for k, v in ctx.prof_get().items() if v[1]}
        print(json.dumps({"kind": kind, "item_chunks": chunks or None, "communicator": "rccl, 1 rank" if chunks else None,
                          "exchange": exchange if chunks else None, "user_side": "unfused" if unfused else "fused",
                          "epoch_ms": round(ms, 3), "kernels_ms": prof, "device_GB": round(ctx.device_bytes() / 1e9, 1)}), flush=True)
        ctx.close()
    comm.close()


if __name__ == "__main__":
    main()
