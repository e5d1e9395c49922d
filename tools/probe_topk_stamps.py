"""Residency timeline of the fused top-k launch.  Builds a DIAGNOSTIC copy of the library with -DPMF_TOPK_STAMPS
(real-time stamps at the begin and end of every wavefront's scan of a user tile; the product library has none), runs
the bench launch (262,144 users x 100,000 items, K = 64) and prints how many wavefronts are inside a scan at 41 instants.
    python tools/probe_topk_stamps.py [k]"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "prob-matrix-factorization_amd")
sys.path[:0] = [ROOT, PKG]
import __graft_entry__ as g  # noqa: E402

work = tempfile.mkdtemp(prefix="topk_diag_")
obj = os.path.join(work, "pmf_topk_diag.o")
lib = os.path.join(work, "libpmf_hip_diag.so")
# (the product library and its objects are built already -- `g.build()` would also LOAD the product library,
#  and this process must load the diagnostic one instead)
base = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-result",
        "-I", os.path.join(ROOT, "include"), "-I", g.CSRC]
subprocess.run(base + g.EXTRA_FLAGS.get("pmf_topk.hip", []) + ["-DPMF_TOPK_STAMPS"] + os.environ.get("PMF_DIAG_DEFS", "").split() + ["-c", os.path.join(g.CSRC, "pmf_topk.hip"), "-o", obj], check=True)
objs = [obj if s == "pmf_topk.hip" else os.path.join(g.CSRC, "obj", s.replace(".hip", ".o")) for s in g.SOURCES]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl"], check=True)
os.environ["PMF_HIP_LIBRARY"] = lib
import pmf_hip  # noqa: E402
from pmf_hip import ARR_FACTOR, ITEM, USER  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
U, I, K, Q = 1_000_000, 100_000, 64, 262_144
rng = np.random.default_rng(0)
users = rng.permutation(U)[:Q].astype(np.int32)
theta = rng.gamma(0.5, 1.0, (U, K))
L = pmf_hip.load()
L.pmf_debug_topk_stamps.argtypes = [C.c_void_p, C.c_int]
for label, beta in (("random item factors", rng.gamma(0.5, 1.0, (I, K))),
                    ("scores descending with the item id", np.outer(np.linspace(2.0, 1.0, I), np.ones(K)))):
    with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
        ctx.set_array(USER, ARR_FACTOR, theta)
        ctx.set_array(ITEM, ARR_FACTOR, beta)
        ctx.topk_items(users, k)
        ctx.prof_enable(True)
        ctx.prof_reset()
        ctx.topk_items(users, k)
        ms, n = ctx.prof_get()["topk"]
        n_waves = Q // 32
        raw = np.zeros(n_waves * 8, dtype=np.int64)
        assert L.pmf_debug_topk_stamps(raw.ctypes.data, n_waves) == 0
        raw = raw.reshape(-1, 8)
        t0 = raw[:, 0].min()
        begin, end = (raw[:, 0] - t0) / 100.0, (raw[:, 1] - t0) / 100.0          # microseconds
        grid = np.linspace(0, end.max(), 41)
        alive = [int(((begin <= t) & (end > t)).sum()) for t in grid]
        print(json.dumps({"items": label, "k": k, "kernel_ms": round(ms / n, 3), "grid_x": int(raw[0, 2]),
                          "waves_in_their_scan_at_41_instants": alive, "span_us": round(float(end.max()), 1),
                          "scan_us_per_user_tile_median": round(float(np.median(end - begin)), 1),
                          "scan_us_per_user_tile_p5_p95": [round(float(np.percentile(end - begin, 5)), 1), round(float(np.percentile(end - begin, 95)), 1)],
                          # shader-clock cycles per wave and scan (3125 tiles), means over the wavefronts
                          "cycles_ranking_candidates_mean": round(float(raw[:, 4].mean())),
                          "cycles_at_stage_barriers_mean": round(float(raw[:, 5].mean())),
                          "candidates_per_wave_scan_mean": round(float(raw[:, 6].mean()), 1),
                          "tiles_with_candidates_mean": round(float(raw[:, 7].mean()), 1),
                          "scan_cycles_at_2.39GHz_median": round(float(np.median(end - begin)) * 2390),
                          # when the (last recorded) scans of the user tiles begin: ms -> number of wavefronts
                          "scan_begin_ms_histogram": {str(b): int(n) for b, n in zip(*np.unique(np.round(begin / 1000.0), return_counts=True))},
                          "xcc_of_blocks_0_256_512_768": [int(raw[(b * 4) % len(raw), 3]) for b in (0, 256, 512, 768)]}), flush=True)
