"""Interleaved A/B of mapping-iteration variants in ONE process (bench.py::mapping_leg's warmed-up step; run-to-run noise of
separate processes is ~2 %, more than the differences looked for): table-gradient clearing policy x Morton key bits.
usage: python tools/ab_mapping.py [iters per block, default 30] [rounds, default 3]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from nicer_slam_amd.fused import tablegrad, render

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
VARIANTS = [("acquire24", dict(inplace=True, consume=False, bits=24)), ("fused24", dict(inplace=True, consume=True, bits=24)),
            ("autograd24", dict(inplace=False, consume=False, bits=24)), ("acquire30", dict(inplace=True, consume=False, bits=30)),
            ("acquire21", dict(inplace=True, consume=False, bits=21))]


def hook(step):
    opt = dict(zip(step.__code__.co_freevars, [c.cell_contents for c in step.__closure__]))["opt"]
    res = {n: [] for n, _ in VARIANTS}
    for rnd in range(ROUNDS):
        for name, v in VARIANTS:
            tablegrad.IN_PLACE, opt.consume_table_grads, render.MORTON_BITS = v["inplace"], v["consume"], v["bits"]
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(ITERS):
                step()
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / ITERS * 1e3)
    return res


with bench.quiet_gc():
    res = bench.mapping_leg(torch.device("cuda", 0), iters=3, cpu=False, step_hook=hook)
for name, ts in res.items():
    print(f"{name:12s} ms/iteration: " + "  ".join(f"{t:7.3f}" for t in ts) + f"   min {min(ts):7.3f}  median {sorted(ts)[len(ts) // 2]:7.3f}")
print(json.dumps(res))
