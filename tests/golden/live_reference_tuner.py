#!/usr/bin/env python3
"""Run THE REFERENCE's `tune_all_models.main()` with one trial per model in the current directory (build container
only; helper of tests/test_live_reference_cpu.py).  Leaves best_hyperparams.txt and stdout_tune.txt."""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, "/root/reference")
from src.experiments import tune_all_models  # noqa: E402

sys.argv = ["tune_all_models", "--n_trials", "1"]
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    tune_all_models.main()
with open("stdout_tune.txt", "w") as fh:
    fh.write(buf.getvalue())
