"""Scratch probe: top-10 throughput at the C2 shape (100k items, K=64)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "prob-matrix-factorization_amd"))
import numpy as np
import pmf_hip
from pmf_hip import USER, ITEM, ARR_FACTOR
U, I, K = 1_000_000, 100_000, 64
rng = np.random.default_rng(0)
ctx = pmf_hip.Context(U, I, K, dtype="f32")
ctx.set_array(USER, ARR_FACTOR, rng.gamma(0.5, 1.0, (U, K)))
ctx.set_array(ITEM, ARR_FACTOR, rng.gamma(0.5, 1.0, (I, K)))
users = np.arange(20_000, dtype=np.int32)
ctx.topk_items(users[:2048], 10)
ctx.prof_enable(True); ctx.prof_reset()
t = time.time(); items, scores = ctx.topk_items(users, 10); dt = time.time() - t
print("20k users top-10 of 100k items:", dt, "s ->", len(users) / dt, "users/s;", ctx.prof_get()["topk"])
flops = 2.0 * len(users) * I * K
print("score GEMM flop", flops / 1e12, "TF")
