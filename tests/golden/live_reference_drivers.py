#!/usr/bin/env python3
"""Run THE REFERENCE's full-training drivers in the current directory (build container only): helper of
tests/test_live_reference_cpu.py::test_driver_outputs_on_disk_equal_the_live_reference.

    cd <dir with data/processed/interactions_*.csv and best_hyperparams.txt>
    python tests/golden/live_reference_drivers.py train+val

The drivers write data/embeddings/<model>/{user,item}_embeddings.csv, config.txt and
data/predictions/<model>/test_predictions.csv below the current directory; this script adds stdout_<model>.txt."""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.path.insert(0, "/root/reference")
from src.experiments.train_gaussian_full import train_full_gaussian  # noqa: E402
from src.experiments.train_hpf_cavi_full import train_full_hpf_cavi  # noqa: E402
from src.experiments.train_hpf_pytorch_full import train_full_hpf_pytorch  # noqa: E402
from src.experiments.train_poisson_full import train_full_poisson  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
for name, fn in (("gaussian_mf", train_full_gaussian), ("poisson_mf", train_full_poisson), ("hpf_cavi", train_full_hpf_cavi),
                 ("hpf_pytorch", train_full_hpf_pytorch)):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        fn(dataset_mode=mode)
    with open(f"stdout_{name}.txt", "w") as fh:
        fh.write(buf.getvalue())
