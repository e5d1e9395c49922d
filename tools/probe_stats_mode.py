"""Per-GPU cost of the multi-GPU code path without the collective: one C2-sized Gaussian (or HPF)
context, fused iteration vs accumulate/finalize iteration over n item chunks.
    python tools/probe_stats_mode.py [gauss|hpf] [chunks ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import pmf_hip  # noqa: E402
from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER, dist as pdist  # noqa: E402
from pmf_hip.synth import BASE_SEED, synth_ratings  # noqa: E402
from helpers import gamma_stats, gauss_stats  # noqa: E402  (caller-owned torch buffers)


class NoComm:
    world = 2
    in_library = False

    def all_reduce(self, t):
        return t

    def all_reduce_async(self, t):
        class W:
            def wait(self):
                pass
        return W()


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "gauss"
    chunk_list = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
    U, I, N, K = 1_000_000, 100_000, 50_000_000, 64
    u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
    dev = torch.device("cuda", 0)
    ctx = pmf_hip.Context(U, I, K)
    rng = np.random.default_rng(1)
    if kind == "gauss":
        ctx.set_ratings(u, i, r - r.mean())
        ctx.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K)))
        ctx.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
        ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
        ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))
        s_item, s_bias = gauss_stats(ctx, dev)

        def step(comm):
            pdist.gaussian_iteration(ctx, comm, s_item, s_bias, 0.5, 1.0, 1.0, 1.0)
    else:
        ctx.set_ratings(u, i, r + 1.0)
        ctx.set_array(USER, ARR_FACTOR, rng.gamma(1.0, 0.3, (U, K)) + 0.1)
        ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(1.0, 0.3, (I, K)) + 0.1)
        ctx.set_array(USER, ARR_PRIOR_RATE, np.full(U, 1.0)); ctx.set_array(ITEM, ARR_PRIOR_RATE, np.full(I, 1.0))
        s_item = gamma_stats(ctx, dev)
        up = ip = (0.3, 0.0, True, 0.3 + K * 0.3, 1.0)

        def step(comm):
            pdist.gamma_iteration(ctx, comm, s_item, up, ip)

    def timed(comm, n=5):
        for _ in range(2):
            step(comm)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            step(comm)
        ctx.sync()
        return (time.perf_counter() - t0) / n * 1e3

    print(f"{kind}: fused single-GPU iteration {timed(None):.2f} ms", flush=True)
    for c in chunk_list:
        ctx.set_row_chunks(ITEM, c)
        print(f"{kind}: accumulate/finalize path, {c} item chunk(s): {timed(NoComm()):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
