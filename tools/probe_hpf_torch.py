import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "prob-matrix-factorization_amd"))
import numpy as np, torch
from src.models.hpf_pytorch import HPF_PyTorch, HPF_PyTorch_Config
U, I, N, K = 11780, 13000, 271000, 10
rng = np.random.default_rng(0)
u = torch.from_numpy(rng.integers(0, U, N)); i = torch.from_numpy(rng.integers(0, I, N))
r = torch.from_numpy(rng.integers(1, 7, N).astype(np.float32))
uc = np.bincount(u.numpy(), minlength=U); ic = np.bincount(i.numpy(), minlength=I)
for dev in ("cuda", "cpu"):
    m = HPF_PyTorch(U, I, uc, ic, HPF_PyTorch_Config(n_factors=K)).to(dev)
    uu, ii, rr = u.to(dev), i.to(dev), r.to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    def step(idx):
        opt.zero_grad(); l = m.loss(uu[idx], ii[idx], rr[idx]); l.backward(); opt.step(); return l
    order = torch.randperm(N, device=dev)
    for k in range(3): step(order[k*4096:(k+1)*4096])
    if dev == "cuda": torch.cuda.synchronize()
    t = time.time()
    for k in range(20): l = step(order[k*4096:(k+1)*4096])
    if dev == "cuda": torch.cuda.synchronize()
    print(dev, "ms/step", (time.time()-t)/20*1e3, flush=True)
    if dev == "cuda":
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for k in range(5): step(order[k*4096:(k+1)*4096])
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
