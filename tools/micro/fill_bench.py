"""Zero fill of a 1 GiB buffer: torch's fill vs nsa_fill_zero.
usage: python tools/micro/fill_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from nicer_slam_amd._native import lib, check

n = 133023682 * 2
buf = torch.empty(n, device="cuda")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_torch = timed(lambda: buf.zero_())
t_nsa = timed(lambda: check(lib.nsa_fill_zero(buf.data_ptr(), n, st)))
print(f"torch fill {t_torch:.1f} us ({n * 4 / t_torch / 1e6:.2f} TB/s)   "
      f"nsa_fill_zero {t_nsa:.1f} us ({n * 4 / t_nsa / 1e6:.2f} TB/s)")
