"""How much of the fused top-k's time is the list-insertion path?  The same launch (262,144 users x 100,000 items,
K = 64, k = 10) on two item tables: random factors (about 92 list changes per user over the scan, i.e. about every
third 32-item tile of a wave finds a candidate) and factors scaled so that every user's scores DESCEND with the item
id (all list changes happen in the first tile; every later tile ends at the threshold compare).
    python tools/probe_topk_inserts.py [--k 1 10 50] [--buffers 0 [1 2]] [--random-only]
--buffers pins the kernel's stage buffering per run (PMF_TOPK_STAGE_BUFFERS; 0 = the library's own choice)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prob-matrix-factorization_amd")]
import pmf_hip  # noqa: E402
from pmf_hip import ARR_BIAS, ARR_FACTOR, ITEM, USER  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, nargs="+", default=[1, 10, 50])
ap.add_argument("--buffers", type=int, nargs="+", default=[0])
ap.add_argument("--random-only", action="store_true")
ap.add_argument("--bias", action="store_true", help="rank under b_u + b_i + dot (the Gaussian models' predict)")
args = ap.parse_args()
U, I, K, Q = 1_000_000, 100_000, 64, 262_144
rng = np.random.default_rng(0)
users = rng.permutation(U)[:Q].astype(np.int32)
theta = rng.gamma(0.5, 1.0, (U, K))
for label, beta in (("random item factors", rng.gamma(0.5, 1.0, (I, K))),
                    ("scores descending with the item id", np.outer(np.linspace(2.0, 1.0, I), np.ones(K))))[:1 if args.random_only else 2]:
  for k in args.k:
    for nb in args.buffers:
        os.environ.pop("PMF_TOPK_STAGE_BUFFERS", None)
        if nb:
            os.environ["PMF_TOPK_STAGE_BUFFERS"] = str(nb)         # read when the context is created
        with pmf_hip.Context(U, I, K, dtype="f32") as ctx:
            ctx.set_array(USER, ARR_FACTOR, theta)
            ctx.set_array(ITEM, ARR_FACTOR, beta)
            if args.bias:
                ctx.set_array(USER, ARR_BIAS, rng.standard_normal(U)); ctx.set_array(ITEM, ARR_BIAS, rng.standard_normal(I))
            ctx.topk_items(users, k, use_bias=int(args.bias))
            ctx.prof_enable(True)
            ctx.prof_reset()
            for _ in range(3):
                ctx.topk_items(users, k, use_bias=int(args.bias))
            ms, n = ctx.prof_get()["topk"]
        tf = 2.0 * Q * I * K / (ms / n * 1e-3) / 1e12
        print(json.dumps({"items": label, "k": k, "bias": args.bias, "stage_buffers": nb or "auto", "kernel_ms": round(ms / n, 3), "TFLOP/s": round(tf, 1),
                          "frac_of_157.3": round(tf / 157.3, 3)}), flush=True)
