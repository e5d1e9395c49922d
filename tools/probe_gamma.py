"""Scratch probe: HPF half-sweep throughput at BASELINE config C3 scale."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "prob-matrix-factorization_amd"))
import numpy as np
import pmf_hip
from pmf_hip import USER, ITEM, ARR_FACTOR, ARR_PRIOR_RATE

U, I, N, K = 1_000_000, 100_000, int(float(sys.argv[1]) if len(sys.argv) > 1 else 50e6), 64
rng = np.random.default_rng(20251226)
t0 = time.time()
u = rng.permutation(U)[np.floor(U * rng.random(N) ** 2.0).astype(np.int64)].astype(np.int32)
i = rng.permutation(I)[np.floor(I * rng.random(N) ** 3.0).astype(np.int64)].astype(np.int32)
x = (rng.choice(6, size=N, p=[0.032, 0.006, 0.012, 0.036, 0.142, 0.772]) + 1).astype(np.float64)
print("gen", time.time() - t0, flush=True)
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
ctx = pmf_hip.Context(U, I, K, dtype=dtype)
t0 = time.time(); ctx.set_ratings(u, i, x); print("set_ratings", time.time() - t0, flush=True)
ctx.set_array(USER, ARR_FACTOR, 0.3 + rng.gamma(1.0, 0.1, size=(U, K)))
ctx.set_array(ITEM, ARR_FACTOR, 0.3 + rng.gamma(1.0, 0.1, size=(I, K)))
ctx.set_array(USER, ARR_PRIOR_RATE, np.ones(U)); ctx.set_array(ITEM, ARR_PRIOR_RATE, np.ones(I))
print("device GB", ctx.device_bytes() / 1e9, flush=True)
def it():
    ctx.gamma_sweep(USER, 0.3, 0.0, True, 5.0 + K * 0.3, 5.0)
    ctx.gamma_sweep(ITEM, 0.3, 0.0, True, 5.0 + K * 0.3, 5.0)
for _ in range(2): it()
ctx.sync(); ctx.prof_enable(True); ctx.prof_reset()
t0 = time.time(); n = 5
for _ in range(n): it()
ctx.sync(); dt = (time.time() - t0) / n
es = 4 if dtype == "f32" else 8
bytes_it = N * (2 * es * K + 16) + (U + I) * 4 * es * K
print(f"iter {dt*1e3:.2f} ms  {N/dt/1e9:.2f} G ratings/s  alg {bytes_it/dt/1e12:.2f} TB/s")
print(ctx.prof_get())
