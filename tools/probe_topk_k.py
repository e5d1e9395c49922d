"""Scratch probe: fused top-k kernel time against k and against the size of the item table (L2 / Infinity-Cache residency)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "prob-matrix-factorization_amd")]
import numpy as np, pmf_hip
from pmf_hip import USER, ITEM, ARR_FACTOR
U, K, Q = 1_000_000, 64, 262144
rng = np.random.default_rng(0)
for I in (100_000, 25_000, 12_500):
    ctx = pmf_hip.Context(U, I, K, dtype="f32")
    ctx.set_array(USER, ARR_FACTOR, rng.gamma(0.5, 1.0, (U, K)))
    ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(0.5, 1.0, (I, K)))
    users = rng.permutation(U)[:Q].astype(np.int32)
    ctx.topk_items(users[:4096], 10)
    for k in (1, 10):
        ctx.prof_enable(True); ctx.prof_reset()
        ctx.topk_items(users, k)
        ms = ctx.prof_get()["topk"][0]
        print(f"I={I} table={I*K*4/1e6:.1f}MB k={k}: {ms:.2f} ms, {2.0*Q*I*K/ms/1e9:.1f} TFLOP/s", flush=True)
    ctx.close()
