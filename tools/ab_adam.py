"""A/B of k_adam_table (nicer_slam_amd.optim.Adam) on a 1 GiB table: us per step and TB/s of its 7 streams.  NSA_LIB_TAG selects a build."""
import os, sys, time
import torch
sys.path.insert(0, ".")
from nicer_slam_amd.optim import Adam
n = 133_000_000 * 2                      # ~ the colour table (1015 MiB)
p = torch.nn.Parameter(torch.randn(n, device="cuda") * 1e-4)
opt = Adam([p], lr=0.01, betas=(0.9, 0.99), eps=1e-15)
p.grad = torch.randn(n, device="cuda") * 1e-3
for _ in range(3):
    opt.step()
torch.cuda.synchronize()
for rnd in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(os.environ.get("NSA_LIB_TAG", "(product)"), f"{us:8.1f} us per step   {7 * n * 4 / (us * 1e-6) / 1e12:.2f} TB/s")
