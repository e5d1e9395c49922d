#!/usr/bin/env python3
"""Run THE REFERENCE's two named drivers' `main()` in the current directory (build container only; helper of
tests/test_live_reference_cpu.py): `train_all_models --dataset_mode train` and `compare_models`.  Their stdout goes
to stdout_train_all.txt / stdout_compare.txt; plots are drawn with the Agg backend and left where the reference
puts them."""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, "/root/reference")
from src.experiments import compare_models, train_all_models  # noqa: E402

for name, fn, argv in (("train_all", train_all_models.main, ["train_all_models", "--dataset_mode", "train"]),
                       ("compare", compare_models.main, ["compare_models"])):
    buf = io.StringIO()
    sys.argv = argv
    with contextlib.redirect_stdout(buf):
        fn()
    with open(f"stdout_{name}.txt", "w") as fh:
        fh.write(buf.getvalue())
