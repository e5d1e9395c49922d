#!/usr/bin/env python3
"""Run THE REFERENCE's `tune_hpf_pytorch.run_tuning()` in the current directory (build container only; helper of
tests/test_live_reference_cpu.py).  Leaves stdout_tune_torch.txt."""
import contextlib
import io
import sys

sys.path.insert(0, "/root/reference")
from src.experiments import tune_hpf_pytorch  # noqa: E402

buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    tune_hpf_pytorch.run_tuning()
with open("stdout_tune_torch.txt", "w") as fh:
    fh.write(buf.getvalue())
