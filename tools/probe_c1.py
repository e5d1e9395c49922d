"""BASELINE config C1 (10k x 2k, 200k ratings, K=16) on the engine: seconds per iteration of the three
CAVI models (the reference's own timings at this size: BASELINE.md section 2)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd")]
import pmf_hip  # noqa: E402
from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER  # noqa: E402
from pmf_hip.synth import BASE_SEED, synth_ratings  # noqa: E402

U, I, N, K = 10_000, 2_000, 200_000, 16
u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
rng = np.random.default_rng(0)
for name in ("hpf", "poisson", "gauss+bias"):
    with pmf_hip.Context(U, I, K) as ctx:
        if name == "gauss+bias":
            ctx.set_ratings(u, i, r - r.mean())
            ctx.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K)))
            ctx.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
            ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
            ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))

            def step():
                ctx.gauss_factor_sweep(USER, 0.3, 0.5); ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
                ctx.gauss_bias_sweep(USER, 0.3, 1.0); ctx.gauss_bias_sweep(ITEM, 0.3, 1.0)
        else:
            ctx.set_ratings(u, i, r + (1.0 if name == "hpf" else 0.0))
            ctx.set_array(USER, ARR_FACTOR, rng.gamma(1.0, 0.3, (U, K)) + 0.1)
            ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(1.0, 0.3, (I, K)) + 0.1)
            hier = name == "hpf"
            if hier:
                ctx.set_array(USER, ARR_PRIOR_RATE, np.ones(U)); ctx.set_array(ITEM, ARR_PRIOR_RATE, np.ones(I))

            def step():
                ctx.gamma_sweep(USER, 0.3, 0.5, hier, 0.3 + K * 0.3, 1.0)
                ctx.gamma_sweep(ITEM, 0.3, 0.5, hier, 0.3 + K * 0.3, 1.0)
        for _ in range(5):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(200):
            step()
        ctx.sync()
        dt = (time.perf_counter() - t0) / 200
        print(f"C1 {name}: {dt * 1e6:.1f} us per iteration = {N / dt:.3e} ratings/s", flush=True)
