"""Per-K epoch times on the C2/C3 data (1M x 100k, 50M ratings): one data generation, then for every K the
HPF-CAVI and / or Gaussian-MF epoch, the dominant kernel's time, the fraction of the section-8(d)
algorithmic-bytes roofline and -- for the Poisson/HPF sweep -- the fraction of the live-measured gather ceiling.

    python tools/k_sweep.py hpf 16 20 32 40 64            # one JSON line per K
    python tools/k_sweep.py gauss 16 30 40 50 64 72 96 128
    (--small: 100k x 10k x 5M)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd")]
import pmf_hip  # noqa: E402
from pmf_hip import ARR_BIAS, ARR_FACTOR, ARR_PRIOR_RATE, ITEM, USER  # noqa: E402
from pmf_hip.synth import BASE_SEED, synth_ratings  # noqa: E402
from bench import HBM_PEAK_GBS, algorithmic_bytes  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    small = "--small" in sys.argv
    kind, ks = args[0], [int(a) for a in args[1:]]
    U, I, N = (100_000, 10_000, 5_000_000) if small else (1_000_000, 100_000, 50_000_000)
    u, i, r = synth_ratings(U, I, N, seed=BASE_SEED)
    steps = 5
    for K in ks:
        ctx = pmf_hip.Context(U, I, K)
        rng = np.random.default_rng(1)
        if kind == "gauss":
            ctx.set_ratings(u, i, r - r.mean())
            ctx.set_array(USER, ARR_FACTOR, 0.1 * rng.standard_normal((U, K)))
            ctx.set_array(ITEM, ARR_FACTOR, 0.1 * rng.standard_normal((I, K)))
            ctx.set_cov_identity(USER); ctx.set_cov_identity(ITEM)
            ctx.set_array(USER, ARR_BIAS, np.zeros(U)); ctx.set_array(ITEM, ARR_BIAS, np.zeros(I))

            def step():
                ctx.gauss_factor_sweep(USER, 0.3, 0.5); ctx.gauss_factor_sweep(ITEM, 0.3, 0.5)
                ctx.gauss_bias_sweep(USER, 0.3, 1.0); ctx.gauss_bias_sweep(ITEM, 0.3, 1.0)
            workload, dominant = "gaussian_mf", "gauss_accum"
        else:
            ctx.set_ratings(u, i, r + 1.0)
            ctx.set_array(USER, ARR_FACTOR, rng.gamma(1.0, 0.3, (U, K)) + 0.1)
            ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(1.0, 0.3, (I, K)) + 0.1)
            ctx.set_array(USER, ARR_PRIOR_RATE, np.full(U, 1.0)); ctx.set_array(ITEM, ARR_PRIOR_RATE, np.full(I, 1.0))
            up = ip = (0.3, 0.0, True, 0.3 + K * 0.3, 1.0)

            def step():
                ctx.gamma_sweep(USER, *up); ctx.gamma_sweep(ITEM, *ip)
            workload, dominant = "hpf_cavi", "gamma_sweep"
        for _ in range(2):
            step()
        ctx.sync()
        ctx.prof_enable(True); ctx.prof_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        prof = ctx.prof_get()
        ctx.prof_enable(False)
        total, dom = algorithmic_bytes(workload, U, I, N, K)
        out = {"kind": kind, "K": K, "kpad": ctx.kpad, "epoch_ms": round(ms, 3), "ratings_per_s": N / ms * 1e3,
               "epoch_frac_of_hbm_algorithmic": total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "kernels_ms": {k: round(v[0] / steps, 3) for k, v in prof.items() if v[1]},
               "dominant_frac_of_hbm_algorithmic": dom / (prof[dominant][0] / steps * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if kind != "gauss":
            cu, ci = ctx.gather_ceiling_ms(USER, 5), ctx.gather_ceiling_ms(ITEM, 5)
            out["gather_ceiling_ms"] = {"user": round(cu, 3), "item": round(ci, 3)}
            out["frac_of_gather_ceiling"] = (cu + ci) / (prof[dominant][0] / steps)
        print(json.dumps(out), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
