"""Run the four full-training drivers in sequence; a failing model is reported
and skipped (reference: src/experiments/train_all_models.py:10-58)."""
import time
import traceback

from src.experiments import _full_training as ft
from src.experiments.train_gaussian_full import train_full_gaussian
from src.experiments.train_hpf_cavi_full import train_full_hpf_cavi
from src.experiments.train_hpf_pytorch_full import train_full_hpf_pytorch
from src.experiments.train_poisson_full import train_full_poisson

STEPS = (("Gaussian MF", train_full_gaussian), ("Poisson MF", train_full_poisson),
         ("HPF (CAVI)", train_full_hpf_cavi), ("HPF (PyTorch)", train_full_hpf_pytorch))


def main():
    mode = ft.mode_argument("Run all full training scripts")
    bar = "=" * 47
    print(f"{bar}\n   RUNNING ALL FULL MODEL TRAINING SCRIPTS\n   Mode: {mode}\n{bar}")
    t0 = time.time()
    for k, (name, fn) in enumerate(STEPS, start=1):
        try:
            print(f"\n\n>>> {k}/{len(STEPS)} Starting {name}...")
            fn(dataset_mode=mode)
        except Exception as exc:  # keep going, as the reference does
            print(f"!!! {name} Failed: {exc}")
            traceback.print_exc()
    print(f"\n{bar}\n   ALL DONE. Total Time: {time.time() - t0:.1f}s\n{bar}")


if __name__ == "__main__":
    main()
