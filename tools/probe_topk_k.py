"""Scratch probe: fused top-k kernel time against the factor count K and the list length k (100k items, 262,144 users)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "prob-matrix-factorization_amd")]
import numpy as np, pmf_hip
from pmf_hip import USER, ITEM, ARR_FACTOR
U, I, Q = 1_000_000, 100_000, 262144
rng = np.random.default_rng(0)
for K in [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]:
    ctx = pmf_hip.Context(U, I, K, dtype="f32")
    ctx.set_array(USER, ARR_FACTOR, rng.gamma(0.5, 1.0, (U, K)))
    ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(0.5, 1.0, (I, K)))
    users = rng.permutation(U)[:Q].astype(np.int32)
    ctx.topk_items(users[:4096], 10)
    for k in (1, 10, 50):
        ctx.prof_enable(True); ctx.prof_reset()
        ctx.topk_items(users, k)
        ms = ctx.prof_get()["topk"][0]
        print(f"K={K} k={k}: {ms:.2f} ms, {Q/ms/1e3:.2f} M users/s, {2.0*Q*I*K/ms/1e9:.1f} TFLOP/s", flush=True)
    ctx.close()
