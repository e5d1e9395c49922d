"""hpf_pytorch training loop at the reference's scale (11,780 x 13,000, 271k ratings, K=10,
batch 4096, 50 epochs: 94.7 s in the reference's plot): eager loop vs HIP-graph replay."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "prob-matrix-factorization_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from src.experiments.train_hpf_pytorch_full import adam_epochs  # noqa: E402
from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config  # noqa: E402

U, I, N, K = 11780, 13000, 271000, 10
rng = np.random.default_rng(0)
un, inn = rng.integers(0, U, N), rng.integers(0, I, N)
dev = torch.device("cuda")
u, i = torch.from_numpy(un).to(dev), torch.from_numpy(inn).to(dev)
r = torch.from_numpy(rng.integers(1, 7, N).astype(np.float32)).to(dev)
uc, ic = np.bincount(un, minlength=U), np.bincount(inn, minlength=I)
for use_graph in (False, True):
    torch.manual_seed(0)
    m = HPF_PyTorch(U, I, uc, ic, HPF_PyTorch_Config(n_factors=K)).to(dev)
    adam_epochs(m, u, i, r, 5e-4, 4096, 1, verbose=False, graph=use_graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    adam_epochs(m, u, i, r, 5e-4, 4096, 50, verbose=False, graph=use_graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"graph={use_graph}: 50 epochs {dt:.2f} s = {dt / (50 * 67) * 1e3:.3f} ms/step", flush=True)
