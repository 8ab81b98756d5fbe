"""bench.py through torch.distributed.run with one rank on the GPU: exercises the RCCL init / barrier / max-over-ranks
path the driver uses for N > 1 (N > 1 itself needs several GPUs; the sharded maths is covered on CPU with gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_single_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"] is not None
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert key in line
    # round 5: the context legs of the default line all ran (no leg swallowed an exception) and no roofline fraction exceeds 1
    pm = line["precision_modes"]
    for key in ("bf16_1024x128", "bf16_512x128", "bf16_colour_1024x192"):
        assert "error" not in pm[key] and pm[key]["ms_per_step"] > 0, pm[key]
    for row in ("pose_only", "pose_only_eager", "pose_only_eager_hip_adam", "faithful"):
        assert "error" not in line["dropin"][row], line["dropin"][row]
    assert 0 < line["roofline"]["frac"] <= 1 and 0 < line["roofline"]["whole_step"]["frac_survey_8d"] <= line["roofline"]["whole_step"]["frac_executed"] <= 1
    if line["colour_gather"] and "frac" in line["colour_gather"]:
        assert 0 < line["colour_gather"]["frac"] <= 1


def test_bench_self_launch_two_ranks_oversubscribed():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run; on a one-GPU box the two ranks
    share device 0 over gloo (smoke mode): the ray-sharded step, the max-over-ranks timing and the labels of an N > 1 line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm-s", "0", "--prewarm-steps", "2",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert cfg["global_rays"] == 2048 and cfg["rays_per_gpu"] == 1024
    assert "2 ranks ray-sharded" in cfg["workload"] and "single MI355X" not in cfg["workload"]
    import torch
    if torch.cuda.device_count() < 2:
        assert cfg["oversubscribed"] is True and cfg["rccl_ranks"] == 0 and "OVERSUBSCRIBED" in cfg["workload"]
    else:
        assert cfg["rccl_ranks"] == 2
    assert line["dropin"] is None and line["mapping_iteration"] is None          # context legs are N = 1 only
