"""Scratch probe: Gaussian ITEM factor half-sweep (gathers 8.3 KB covariance rows of the USER table) against the size
of that table -- does an Infinity-Cache-resident user block make the item side faster than DRAM streaming?
(The upper bound of what cache-blocking the item sweep over user ranges could buy.)"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "prob-matrix-factorization_amd")]
import pmf_hip
from pmf_hip import ARR_BIAS, ARR_FACTOR, ITEM, USER
from pmf_hip.synth import BASE_SEED, synth_ratings
K, I, N = 64, 100_000, 50_000_000
for U in (1_000_000, 120_000, 60_000, 30_000, 15_000):
    u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
    ctx = pmf_hip.Context(U, I, K)
    rng = np.random.default_rng(1)
    ctx.set_ratings(u, i, r - r.mean())
    ctx.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K))); ctx.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
    ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
    ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))
    ctx.gauss_factor_sweep(USER, 0.3, 0.5); ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
    ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
    for _ in range(4):
        ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
    ctx.sync()
    ms = ctx.prof_get()["gauss_accum"][0] / 4
    kp = K * (K + 1) // 2
    gb = N * (4 * kp + 4 * K + 12) / 1e9
    print(json.dumps({"U": U, "user_cov_table_MB": U * kp * 4 / 1e6, "item_side_ms": round(ms, 2), "algorithmic_TBps": round(gb / ms, 2)}), flush=True)
    ctx.close()
