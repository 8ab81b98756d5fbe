#!/usr/bin/env python
"""bench.py -- rays/s of one NICER-SLAM *tracking iteration*  on MI355X.

One "step" = one pass of the hot path over one synthetic batch: camera 7-vector -> c2w -> rays -> hierarchical
sampler (E=640 SDF evaluations/ray) -> S=128 composite samples/ray through the three grid encoders + SDF/colour MLPs
-> SDF->density composite -> L1(rgb) -> backward to the pose gradient -> Adam step on the camera.  Real shipped
network/grid sizes (1 GiB colour table), fp32, synthetic inputs already resident in HBM when the clock starts.
N > 1: every rank renders its own 1024-ray shard (rays are independent given replicated parameters) and the only
exchange is one fused 9-float RCCL all-reduce (pose gradient, loss, ray count) per step  ->  "scaling": "weak".

    python bench.py [--gpus N --steps K --warmup W]
N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment) or directly as `python bench.py --gpus N`, which re-executes itself under torch.distributed.run on
127.0.0.1.  `--global-rays G` fixes the TOTAL batch (G/N rays per GPU) -> "scaling": "strong"; default: 1024 rays per GPU.
Prints ONE JSON line on rank 0.
"""
import argparse
import contextlib
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured streaming copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak (MI355X_MICROARCH.md); only used with --precision bf16*
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak = the peak for this config's dtype (MI355X_MICROARCH.md)

# Algorithmic multiply-accumulates per point of each per-point kernel (true layer sizes, no padding):
#   SDF nets 71->64(->64->64)->65 Softplus, colour net 129->64->64->3; forward kernels include the reverse pass that
#   yields grad sdf, backward kernels include the recomputation + tangent sweep + reverse sweep.
# Matrix instructions per 32 points in the THREE-piece operand form (static: groups x tiles x 6 products of v_mfma_f32_32x32x16_bf16,
# see csrc/): with 32 cycles per instruction per SIMD this gives the matrix-pipe busy time, reported beside the fp32-equivalent `frac`.
# The quad tiling issues twice as many 16x16x32 instructions of half the duration -- the same busy cycles
# (profiles/r02_pmc_per_kernel_*.csv).  The library's operand form (csrc/mlp_common.hpp::NSA_FORM, asked at run time:
# nsa_operand_form) scales the count: form 2 = two fp16 pieces per operand, FOUR v_mfma_f32_32x32x16_f16 per block (x 4 / 6).
MFMA_PER_TILE = {"k_sampler_sdf": 216, "k_sdfnet_fwd<coarse>": 180, "k_sdfnet_fwd<fine>": 372, "k_sdfnet_fwd<pair>": 552, "k_sdfnet_bwd<coarse>": 288,
                 "k_sdfnet_bwd<fine>": 672, "k_colour_fwd": 156, "k_colour_bwd": 168, "k_colour_coarse_bwd": 168 + 288}
# (round 5: the data-path colour backward no longer recomputes its forward -- ReLU masks and sigmoid outputs come from the save area --
#  so its count is the two reverse GEMMs only: 12 544 MAC, 168 MFMAs per 32 points; round 4: 25 088 / 324)
N_SIMD, NOMINAL_GHZ = 1024, 2.4

ALGO_MAC = {
    "k_sampler_sdf": 17408,            # coarse (4544+64) + fine (4544+2*4096+64): sdf rows only
    "k_sdfnet_fwd<coarse>": 13248, "k_sdfnet_fwd<fine>": 29632, "k_sdfnet_fwd<pair>": 13248 + 29632,
    "k_sdfnet_bwd<coarse>": 22400, "k_sdfnet_bwd<fine>": 55168,
    "k_colour_fwd": 12544, "k_colour_bwd": 12544, "k_colour_coarse_bwd": 12544 + 22400,
}


@contextlib.contextmanager
def quiet_gc():
    """Timed regions run with Python's cyclic collector off, after a full collection -- as `timeit` does.  A generation-2 pass over
    this process's ~10^6 objects is a 35-60 ms host stall (measured: profiles/r03_ab_experiments.txt r3ab), i.e. 50-100
    tracking iterations; with K = 20 timed steps one such pause multiplies the measured step time by 3-5."""
    was = gc.isenabled()
    gc.collect()          # (callers enter BEFORE their warm-up steps: the collection itself idles the GPU for tens of ms, and the
    gc.disable()          #  first ~20 iterations after an idle gap run at lower clocks)
    try:
        yield
    finally:
        if was:
            gc.enable()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="seconds of an unrelated GEMM loop before the W warm-up steps: a step is < 1 ms, so W steps alone "
                         "end before the GPU's power management has left its idle clocks (measured: 1.1 vs 0.8 ms/step)")
    ap.add_argument("--prewarm-steps", type=int, default=100,
                    help="iterations of the tracker itself after the GEMM loop, followed by a reset of camera / optimizer state to "
                         "their initial values: the first ~20 iterations after construction run ~4 %% slower (cold TLBs / Infinity "
                         "Cache for the 1 GiB table; profiles/r03_ab_experiments.txt r3ab) -- with the driver's K = 20 that IS the sample")
    ap.add_argument("--rays", type=int, default=1024, help="rays per GPU per step (weak scaling)")
    ap.add_argument("--global-rays", type=int, default=0,
                    help="total rays per step over all GPUs (strong scaling: each rank renders global/N); e.g. 4096 = BASELINE "
                         "configs[2], 1024 = configs[1] spread over N GPUs")
    ap.add_argument("--no-dropin", action="store_true", help="skip the SLAMNetwork.forward + autograd leg (N = 1 only)")
    ap.add_argument("--samples", type=int, default=128, help="composite samples per ray (N_samples = S-34)")
    ap.add_argument("--engine", default="auto", choices=["auto", "fused", "composed"])
    ap.add_argument("--param-grads", action="store_true",
                    help="also produce (and discard) table/MLP gradients like the reference's tracking loop")
    ap.add_argument("--chunks", type=int, default=1,
                    help="KernelTracker: independent ray chunks on their own HIP streams inside the graph (fork / join)")
    ap.add_argument("--graph-collective", action="store_true",
                    help="N > 1: capture the 9-float all-reduce and the Adam step into the hipGraph too (default: env "
                         "NSA_GRAPH_COLLECTIVE; off)")
    ap.add_argument("--no-graph", action="store_true", help="launch every op eagerly instead of replaying a hipGraph")
    ap.add_argument("--autograd", action="store_true",
                    help="drive the model through torch autograd (TrackingStepper) instead of the kernel sequence")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16_colour"],
                    help="MLP GEMM operands: fp32 = the reference's precision (default, the BASELINE metric); bf16 / "
                         "bf16_colour = the optional reduced-precision modes of BASELINE configs[2]/[4] (NOT the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precision-modes", action="store_true", help="skip the bf16 / bf16_colour context rows (N = 1 only)")
    ap.add_argument("--no-mapping", action="store_true", help="skip the (untimed-for-value) mapping-iteration leg")
    ap.add_argument("--no-small-shapes", action="store_true", help="skip the strong-scaling-tail rows (128 / 256 / 512 rays; N = 1 only)")
    ap.add_argument("--cpu-rays", type=int, default=1024)
    ap.add_argument("--cpu-all-cores-rays", type=int, default=0,
                    help="rays of a LIVE second CPU figure taken with torch.set_num_threads(os.cpu_count()) (BASELINE.md section 3); default 0: "
                         "quote the committed measurement profiles/rNN_cpu_all_cores.json instead (one iteration costs ~48 s with 256 threads)")
    ap.add_argument("--only-mapping", type=int, default=0, metavar="ITERS",
                    help="profiling aid: run ONLY the mapping-iteration leg with this many timed iterations and print its dict")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 4],
                    help="BASELINE.json configs[i] as a preset: 1 = 1024 rays x 128 samples, one GPU, fp32 (the default); "
                         "2 = --gpus 8 --global-rays 4096 --precision bf16; 4 = --gpus 8 --global-rays 8192 --samples 192 "
                         "--precision bf16_colour (bf16 MLPs with the SDF head in fp32).  Explicit flags given with it must agree.")
    args = ap.parse_args()
    preset = {1: dict(gpus=1, global_rays=0, samples=128, precision="fp32"),
              2: dict(gpus=8, global_rays=4096, samples=128, precision="bf16"),
              4: dict(gpus=8, global_rays=8192, samples=192, precision="bf16_colour")}.get(args.config)
    if preset:
        defaults = dict(gpus=1, global_rays=0, samples=128, precision="fp32")
        for k, v in preset.items():
            if getattr(args, k) != defaults[k] and getattr(args, k) != v:
                ap.error(f"--config {args.config} means --{k.replace('_', '-')} {v}, got {getattr(args, k)}")
            setattr(args, k, v)
    return args


def workload_label(args, world, oversub):
    """config.workload from the arguments actually run (never a fixed string), and the BASELINE.json config it is, if any."""
    g_rays = args.rays * world
    prec = {"fp32": "fp32", "bf16": "bf16 MLP operands", "bf16_colour": "bf16 colour MLP + fp32 SDF head"}[args.precision]
    where = ("single MI355X" if world == 1 else
             f"{world} ranks ray-sharded ({args.rays} rays/rank, {'strong' if args.global_rays else 'weak'} scaling), one 9-float "
             + ("gloo all-reduce per step, ranks OVERSUBSCRIBED on fewer GPUs (smoke mode, not a scaling number)" if oversub
                else "RCCL all-reduce per step, one MI355X per rank"))
    which = "no BASELINE config (custom shape)"
    if world == 1 and args.rays == 1024 and args.samples == 128 and args.precision == "fp32":
        which = "BASELINE configs[1]"
    elif world == 8 and g_rays == 4096 and args.samples == 128 and args.precision == "bf16":
        which = "BASELINE configs[2]"
    elif world == 8 and g_rays == 8192 and args.samples == 192 and args.precision == "bf16_colour":
        which = "BASELINE configs[4] (sample counts; synthetic Replica-sized scene)"
    elif args.rays == 1024 and args.samples == 128 and args.precision == "fp32" and not args.global_rays:
        which = f"BASELINE configs[1] per GPU x {world} (weak scaling of the metric's 1/2/4/8-GPU row)"
    elif g_rays == 1024 and args.samples == 128 and args.precision == "fp32":
        which = f"BASELINE configs[1] spread over {world} GPUs (strong scaling)"
    return (f"Replica room0 tracking iteration, {g_rays} rays x {args.samples} samples (+640 sampler evaluations/ray), {where}, "
            f"{prec} [{which}]")


class DS:
    img_res = (680, 1200)


def make_model(args, device):
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    conf = replica_model_conf(n_samples=args.samples - 34, n_samples_eval=640, n_samples_extra=32, use_warp_loss=False)
    model = SLAMNetwork(conf, dataset=DS(), n_images=2000).to(device)
    model.train()
    model.engine = args.engine
    model.mlp_precision = args.precision
    if not args.param_grads:
        for p in model.parameters():
            p.requires_grad_(False)
    return model, conf


def synth_batch(gen, n_rays, device):
    """SURVEY.md 8d synthetic inputs: 680x1200 image, K=(600,600,599.5,339.5), random pixels, U(0,1) colours."""
    H, W = DS.img_res
    idx = torch.randint(H * W, (1, n_rays), generator=gen, device=device)
    uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    return uv, torch.rand(n_rays, 3, generator=gen, device=device)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.only_mapping:
        self_launch(args)                # (the ranks it starts inherit this process's stdout untouched)
    # stdout carries ONE JSON line and nothing else: native libraries write to file descriptor 1 behind Python's back (RCCL prints a
    # version banner when a communicator is created) -- descriptor 1 is pointed at stderr for the whole run and the line goes to a
    # private duplicate of the original stdout
    sys.stdout.flush()
    out_fd = os.dup(1)
    os.dup2(2, 1)
    global _emit
    def _emit(text):
        sys.stdout.flush()
        os.write(out_fd, (text + "\n").encode())
    if args.only_mapping:
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        _emit(json.dumps(mapping_leg(torch.device("cuda", 0), iters=args.only_mapping, cpu=not args.no_cpu_baseline)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    n_dev = torch.cuda.device_count()
    oversub = world > n_dev          # smoke-test mode: more ranks than GPUs (RCCL refuses two ranks on one device)
    torch.cuda.set_device(local % n_dev)
    device = torch.device("cuda", local % n_dev)
    if world > 1:
        if oversub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    if args.global_rays:
        assert args.global_rays % world == 0, "--global-rays must be divisible by --gpus"
        args.rays = args.global_rays // world

    from nicer_slam_amd.hashencoder import backend as be
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    from nicer_slam_amd.fused.pack import operand_form as _operand_form
    operand_form = _operand_form()
    model, conf = make_model(args, device)
    K = torch.eye(4, device=device)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None]
    gen = torch.Generator(device=device).manual_seed(1 + rank)
    total = args.warmup + args.steps
    batches = [synth_batch(gen, args.rays, device) for _ in range(total)]
    cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device)
    cam = cam + 1e-3 * torch.randn(7, device=device, generator=gen)
    if world > 1:
        dist.broadcast(cam, 0)
    from nicer_slam_amd.tracking import TrackingStepper, KernelTracker
    use_graph = not args.no_graph and not args.param_grads
    Stepper = TrackingStepper if (args.autograd or args.param_grads) else KernelTracker
    extra = {"chunks": args.chunks, "graph_collective": True if args.graph_collective else None} if Stepper is KernelTracker else {}
    stepper = Stepper(model, K, args.rays, cam, lr=0.005, use_graph=use_graph, world=world, **extra)
    ray_chunks = getattr(stepper, "chunks", 1)
    collective_in_graph = bool(getattr(stepper, "collective_in_graph", False))

    def step(i):
        uv, gt = batches[i]
        return stepper.step(uv, gt)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gc_off = quiet_gc()
    gc_off.__enter__()              # collector off from here to the end of the timed region (prewarm and warm-up included)
    probe_tflops = None
    if args.prewarm_s > 0:          # bring the clocks up with work that touches none of the tracker's state or caches
        a = torch.randn(4096, 4096, device=device)
        t_pre, n_mm = time.perf_counter(), 0
        while time.perf_counter() - t_pre < args.prewarm_s:
            for _ in range(8):
                a @ a
            torch.cuda.synchronize()
            n_mm += 8
        # the same loop doubles as a probe of the box's clock / power state (library fp32 GEMM rate): boxes of this pool differ
        # by up to 1.3x with identical binaries, and every kernel of the iteration scales with it
        probe_tflops = round(n_mm * 2 * 4096 ** 3 / (time.perf_counter() - t_pre) / 1e12, 1)
        del a
    cache_prewarm = 0
    if args.prewarm_steps > 0 and hasattr(stepper, "reset"):
        # ... and the memory side with the iteration itself; afterwards camera, Adam moments, step counter and candidate are put
        # back, so the W + K steps below start from the state they would have started from without this
        # (its own batches, drawn from a separate generator: none of the W + K batches below has been rendered before it is timed)
        gen_pre = torch.Generator(device=device).manual_seed(1000 + rank)
        pre_batches = [synth_batch(gen_pre, args.rays, device) for _ in range(min(args.prewarm_steps, 32))]
        cam_start = stepper.cam.detach().clone()
        for i in range(args.prewarm_steps):
            stepper.step(*pre_batches[i % len(pre_batches)])
        stepper.reset(cam_start)
        del pre_batches
        cache_prewarm = args.prewarm_steps
    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        last = step(i)
    fence()
    dt = time.perf_counter() - t0
    gc_off.__exit__(None, None, None)
    last = float(last)
    # Per-kernel durations: graph nodes cannot be bracketed by events, so the same K batches are run once more,
    # eagerly, right after the timed region with an event pair around every launch of ours (on the launch stream).
    eager = stepper if not use_graph else Stepper(model, K, args.rays, stepper.cam.detach(), lr=0.005,
                                                  use_graph=False, world=1)
    be.PROFILE = []
    for i in range(args.warmup, total):
        eager.step(*batches[i])
    torch.cuda.synchronize()
    prof, be.PROFILE = be.PROFILE, None
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    rccl_ranks = 0
    per_rank_ms, exchange_us = None, None
    if world > 1:
        # what the N > 1 line needs to explain itself: every rank's own time for the K steps (the reported one is their max) and
        # the cost of the step's only exchange -- the 9-float all-reduce + nsa_adam_step_scaled -- measured alone, back to back
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank_ms = [round(float(x.item()) / args.steps * 1e3, 4) for x in every]
        if hasattr(stepper, "_exchange"):
            fence()
            t_ex = time.perf_counter()
            for _ in range(50):
                stepper._exchange()
            fence()
            exchange_us = round((time.perf_counter() - t_ex) / 50 * 1e6, 1)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                      # the number of ranks that actually took part in a collective
        rccl_ranks = 0 if oversub else int(ones.item())
        assert int(ones.item()) == dist.get_world_size() == world
    dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        rays_total = args.rays * world * args.steps
        # dominant kernel of OUR kernels over the timed region, from events on the launch stream
        agg = {}
        for name, nbytes, e0, e1 in prof:
            a = agg.setdefault(name, [0.0, 0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += nbytes
            a[2] += 1
        roof = None
        if agg:
            name, (tms, nbytes, n) = max(agg.items(), key=lambda kv: kv[1][0])
            pts = args.rays * (640 if name == "k_sampler_sdf" else args.samples)
            # HBM bytes per launch of that kernel: FETCH_SIZE + WRITE_SIZE from separate `rocprofv3 --pmc` passes over this same
            # command (tools/profile_round.sh -> profiles/rNN_hbm_traffic.json, newest round wins); counters cannot be read
            # from inside the run, so the value is null when no committed summary names this kernel.
            traffic, traffic_src = None, None
            import glob
            for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
                t = json.load(open(tpath)).get(name)
                if t is not None:
                    traffic, traffic_src = t, os.path.relpath(tpath, ROOT)
                    break
            if name in ALGO_MAC:     # per-point MLP kernels: bounded by the matrix pipe
                flops = 2.0 * ALGO_MAC[name] * pts
                ach = flops / (tms / n * 1e-3) / 1e12
                bf16_kernel = args.precision == "bf16" or (args.precision == "bf16_colour" and "colour" in name)
                peak = MFMA_BF16_PEAK_TFLOPS if bf16_kernel else MFMA_F32_PEAK_TFLOPS
                roof = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                        "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "flops_per_launch": flops,
                        "operand_form": None if bf16_kernel else operand_form,
                        "mfma_path": "plain bf16 operands, one MFMA per product block (dense bf16 peak)" if bf16_kernel else
                        ("fp32 products as 4 fp16 MFMAs on operands split into two round-to-nearest fp16 pieces after a per-point "
                         "power-of-two scaling (each operand held to 2^-23, the four products exact; against float64 as close as the three-piece "
                         "form and closer than an fp32 evaluation, tests/test_operand_form_{cpu,gpu}.py, DESIGN 4.4); achieved counts "
                         "algorithmic fp32 flops once" if operand_form == 2 else
                         "fp32 products as 6 bf16 MFMAs on 3-way split operands (fp32-faithful); achieved counts "
                         "algorithmic fp32 flops once")}
            else:
                ach = nbytes / (tms * 1e-3) / 1e9
                roof = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                        "bytes_per_launch": nbytes // n}
            if name in MFMA_PER_TILE and args.precision == "fp32":
                tiles = (pts + 31) // 32
                per_tile = MFMA_PER_TILE[name] * (4 if operand_form == 2 else 6) // 6
                busy = per_tile * tiles * 32.0 / N_SIMD                           # matrix-pipe busy cycles per SIMD
                roof["mfma_pipe"] = {"instructions_per_launch": per_tile * tiles, "busy_cycles_per_simd": round(busy),
                                     "utilisation_at_2.4GHz": round(busy / (tms / n * 1e-3 * NOMINAL_GHZ * 1e9), 4),
                                     "note": ("4 fp16" if operand_form == 2 else "6 bf16") + " MFMAs per fp32 product block; the "
                                             "fp32-equivalent frac above prices the kernel against the fp32 peak, this against the "
                                             "matrix pipe it actually uses"}
            roof.update({"launches": n, "avg_launch_us": round(tms / n * 1e3, 2), "share_of_step": round(tms / n / ms, 4),
                         "all_kernels_us": {k: round(v[0] / v[2] * 1e3, 1) for k, v in sorted(agg.items())},
                         "all_kernels_us_note": "event pairs around every launch of an EAGER replay of the timed batches (graph nodes "
                                                "cannot be bracketed): eager launches run 1-2 % above their in-graph time, so the sum "
                                                "may exceed ms_per_step"})
            if args.precision == "fp32" and args.engine != "composed":
                # whole iteration against the fp32 matrix line, two counts: the work the kernels EXECUTE (backward kernels
                # recompute their forward: ALGO_MAC includes it) and SURVEY 8d's algorithmic count (127 488 MAC per composite
                # point, data gradients only, + the sampler's sdf-only evaluation 17 408 MAC per sampler point)
                P_c, P_s = args.rays * args.samples, args.rays * 640
                executed = 2.0 * sum(ALGO_MAC[k] * (P_s if k == "k_sampler_sdf" else P_c) for k in agg if k in ALGO_MAC)
                survey = 2.0 * (127488 * P_c + 17408 * P_s)
                roof["whole_step"] = {"executed_gflop": round(executed / 1e9, 2), "survey_8d_gflop": round(survey / 1e9, 2),
                                      "frac_executed": round(executed / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                      "frac_survey_8d": round(survey / (ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                      "note": "of the fp32 matrix line (157.3 TFLOP/s), per GPU"}
        # (the CPU figure is taken on rank 0 of the ONE-GPU run only: with N > 1 the other ranks would sit in the communicator's tear-down
        #  for its 13 s)
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(args, model, conf)
        mapping, dropin, ref_gpu, prec_modes, small = None, None, None, None, None
        gather = colour_gather_roofline(agg, args) if agg else None
        if world == 1 and not args.no_precision_modes and args.engine != "composed" and args.precision == "fp32":
            prec_modes = precision_modes_leg(args, device, K)
        if (world == 1 and not args.no_small_shapes and args.engine != "composed" and args.precision == "fp32" and args.rays == 1024
                and args.samples == 128):
            small = small_shapes_leg(args, device, K, ms)
        if world == 1 and not args.no_dropin and args.engine != "composed" and args.precision == "fp32":
            dropin = dropin_leg(args, device, K, batches)
            ref_gpu = reference_shaped_gpu_leg(args, device, K, batches, rays_total / dt)
        if world == 1 and not args.no_mapping and args.engine != "composed" and args.precision == "fp32":
            del stepper, eager
            mapping = mapping_leg(device)
        line = {
            "metric": "rays/sec (fwd+bwd), one tracking iteration", "value": round(rays_total / dt, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "strong" if args.global_rays else "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 MLP operands (f32 accumulate, encoders and compositing f32)",
                      "bf16_colour": "bf16 colour-MLP operands, f32 SDF head"}[args.precision], "data": "synthetic",
            "config": {"workload": workload_label(args, world, oversub), "baseline_config_preset": args.config or None,
                       "rays_per_gpu": args.rays, "samples_per_ray": args.samples, "sampler_evals_per_ray": 640,
                       "global_rays": args.rays * world,
                       "engine": "fused" if Stepper.__name__ == "KernelTracker" else model.last_engine, "param_grads": args.param_grads,
                       "hip_graph": bool(use_graph), "driver": Stepper.__name__, "python_gc": "off inside the timed regions (as timeit)", "ray_chunks": ray_chunks, "clock_prewarm_s": args.prewarm_s, "cache_prewarm_steps": cache_prewarm, "cache_prewarm_batches": "disjoint from the warm-up and timed batches", "box_probe_fp32_gemm_tflops": probe_tflops,
                       "parallelism": f"ray-shard x{world}" if world > 1 else "single",
                       "rccl_ranks": rccl_ranks,
                       "exchange": None if world == 1 else "one 9-float all-reduce (pose gradient, loss, ray count) per step"
                                   + (", captured in the hipGraph" if collective_in_graph else ", between graph replay and Adam launch"),
                       "per_rank_ms_per_step": per_rank_ms,
                       "exchange_us_per_step": exchange_us if world > 1 else None,
                       "exchange_measured": None if world == 1 else "all-reduce of the 9-float message + nsa_adam_step_scaled, 50 back-to-back "
                                            "calls after the timed region (host-issued: includes launch latency, which a step hides "
                                            "behind the graph replay of the next iteration only when collective_in_graph)",
                       "oversubscribed": oversub or None},
            "final_loss": round(last, 6),
            "roofline": roof, "colour_gather": gather, "cpu_baseline": cpu, "reference_shaped_gpu": ref_gpu, "dropin": dropin,
            "mapping_iteration": mapping, "precision_modes": prec_modes, "small_shapes": small,
        }
        _emit(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def mapping_signature(rays=8192, frames=8, samples=98):
    """What a committed atomic-request profile (profiles/rNN_mapping_pmc_per_kernel.csv) must have been taken with to be quoted
    beside a launch time of THIS run: the batch shape and the kernels' tilings (tools/profile_mapping.sh stores it next to the CSV)."""
    from nicer_slam_amd.fused.sampler import DEFAULT_TILES
    from nicer_slam_amd.fused.render import MORTON_BITS
    return {"rays": rays, "keyframes": frames, "samples_per_ray": samples, "tiles": dict(sorted(DEFAULT_TILES.items())),
            "morton_bits": MORTON_BITS}      # (the launch order decides how many rows merge into one atomic request)


def mapping_leg(device, rays=8192, frames=8, iters=5, cpu=True, step_hook=None):
    """Context number, not `value`: one MAPPING iteration as the shipped Replica configuration runs it
    (code/confs/replica/runconf_replica_1.conf; volsdf_train.py:548-576): SLAMNetwork.forward(mode="mapping", stage "fine",
    colour stage "highfreq", use_warp_loss = true, mapping_patchsizes = [1], flow edges between neighbouring keyframes) ->
    SLAMLoss with that conf's weights (rgb L1, eikonal 0.1, smooth 0.005, ssi depth 0.1, normal L1/cos 0.05, patch warp 0.5,
    flow 0.001) -> backward to the three tables + coarse SDF MLP + colour MLP -> Adam (lr x20 / x20 / x5 for the grids,
    volsdf_train.py:150-174).  8192 rays over 8 keyframes, 98 samples/ray, 180 k eikonal points; frames resident in HBM
    (feed.py); fused engine, nicer_slam_amd.optim.Adam."""
    from nicer_slam_amd.feed import FrameFeed
    from nicer_slam_amd.model.loss import SLAMLoss
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.optim import Adam
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(64, 640, 32, use_warp_loss=True, mapping_patchsizes=[1]), dataset=DS(),
                        n_images=2000).to(device)
    model.train().freeze_fine_mlp()
    groups = [{"params": list(model.implicit_network.fine.grid_parameters()), "lr": 0.04},
              {"params": list(model.implicit_network.coarse.grid_parameters()), "lr": 0.04},
              {"params": list(model.rendering_network.grid_parameters()), "lr": 0.01},
              {"params": list(model.rendering_network.mlp_parameters()), "lr": 0.002},
              {"params": list(model.implicit_network.coarse.mlp_parameters()), "lr": 0.002}]
    opt = Adam(groups, betas=(0.9, 0.99), eps=1e-15)

    class TrainDS:
        data_dir = "../Datasets/processed/Replica"
    crit = SLAMLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, train_dataset=TrainDS(), scan_id=1,
                    assign_scale_shift_init=True, smooth_weight=0.005, warp_loss_type="l1", depth_weight=0.1,
                    normal_l1_weight=0.05, normal_cos_weight=0.05, flow_weight=0.001, warp_loss_weight=0.5)
    g = torch.Generator(device=device).manual_seed(1)
    K = torch.eye(4, device=device)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    cams = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device).repeat(frames, 1)
    cams = cams + 0.01 * torch.randn(frames, 7, device=device, generator=g)
    H, W = DS.img_res
    feed = FrameFeed((H, W), device=device, capacity=frames)
    key_ids = [10 * i for i in range(frames)]
    with torch.no_grad():
        poses = get_camera_from_tensor(cams)
    for i, fid in enumerate(key_ids):                       # synthetic frames, uploaded once (smooth depth, random colour)
        feed.add_frame(fid, rgb=torch.rand(H * W, 3, device=device, generator=g),
                       depth=0.02 + 0.01 * torch.rand(H * W, 1, device=device, generator=g),
                       normal=torch.nn.functional.normalize(torch.randn(H * W, 3, device=device, generator=g), dim=-1),
                       gt_depth=1.0 + 0.5 * torch.rand(H * W, 1, device=device, generator=g), intrinsics=K, pose=poses[i])
    # flow graph as build_graph makes it (volsdf_train.py:312-324): keyframes at most 30 frames apart, both directions
    pairs = [(a, b) for a in range(frames) for b in range(frames) if a != b and abs(key_ids[a] - key_ids[b]) <= 30]
    idii = torch.tensor([a for a, _ in pairs], device=device)
    idjj = torch.tensor([b for _, b in pairs], device=device)
    edges = (idii, idjj, None, None)
    n = rays // frames
    flow_images = (torch.randn(len(pairs), H * W, 2, device=device, generator=g) * 5,
                   torch.rand(len(pairs), H * W, device=device, generator=g) > 0.2)

    def step():
        sel = feed.change_sampling_idx(n, generator=g)
        indices, inp, gt = feed.batch(key_ids, full="store")
        gt["edges"] = edges
        # select_flow_uv (volsdf_train.py:348-361): the GT flow of the sampled pixels of every edge
        gt["flow"], gt["flow_mask"] = flow_images[0].index_select(1, sel), flow_images[1].index_select(1, sel)
        opt.zero_grad()
        out = model(inp, indices.to(device), gt, keyframe_list=key_ids, mode="mapping", stage="fine", color_stage="highfreq",
                    frame_idx=key_ids[-1])
        loss = crit(out, gt, key_ids, frame_idx=key_ids[-1], stage="fine")["loss"]
        loss.backward()
        opt.step()
        return loss

    with quiet_gc():
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            last = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    assert model.last_engine == "fused"
    if step_hook is not None:          # tools/profile_mapping_host.py: the warmed-up iteration, handed out for a profiler
        return step_hook(step)
    # per-kernel durations of four more iterations (event pairs on the launch stream; means per iteration -- a single iteration's
    # reading of the dominant kernel moved by 10 % between runs) and the roofline of the dominant kernel
    import nicer_slam_amd.hashencoder.backend as be
    PROF_ITERS = 4
    be.PROFILE = []
    for _ in range(PROF_ITERS):
        step()
    torch.cuda.synchronize()
    prof, be.PROFILE = be.PROFILE, None
    agg = {}
    for name, _, e0, e1 in prof:
        agg[name] = agg.get(name, 0.0) + e0.elapsed_time(e1) / PROF_ITERS
    S = 98
    P = rays * S
    roof = None
    if agg:
        name, tms = max(agg.items(), key=lambda kv: kv[1])
        # algorithmic bytes per composite point of the MAP backward kernels: saved forward state read + table rows
        # added (2^3 corners x levels x C floats, counted once) + emission rows written for the weight-gradient GEMMs
        per_point = {"k_colour_bwd<map>": (512 + 16 * 8 * 2 * 4 + 389 * 4, 16 * 8),
                     "k_sdfnet_bwd<fine,map>": (3 * 8 * 8 * 4 * 4 + 8 * 8 * 4 * 4, 8 * 8),
                     "k_sdfnet_bwd<coarse,map>": (3 * 4 * 8 * 8 * 4 + 4 * 8 * 8 * 4 + 464 * 4, 4 * 8)}.get(name)
        if per_point:
            nbytes, rows = per_point[0] * P, per_point[1] * P
            ach = nbytes / (tms * 1e-3) / 1e9
            # The binding quantity of the MAP kernels is the ATOMIC REQUEST rate, not bytes: fp32 atomics retire at ~20 G requests/s
            # chip-wide on MI355X whatever the table size or contention (tools/micro/atomic_bench.hip; a request = the lanes of one
            # instruction that fall into one 64-byte segment).  Requests per launch come from the TCC_ATOMIC counter of a separate
            # `rocprofv3 --pmc` pass over this same leg (tools/profile_mapping.sh -> profiles/rNN_mapping_pmc_per_kernel.csv;
            # counters cannot be read from inside the run, so the newest committed summary is quoted and named).
            pmc_key = {"k_colour_bwd<map>": "k_colour_bwd<true>", "k_sdfnet_bwd<fine,map>": "k_sdfnet4_bwd<8, 4, 3, true>",
                       "k_sdfnet_bwd<coarse,map>": "k_sdfnet4_bwd<4, 8, 1, true>"}[name]
            reqs, req_src, req_ok = None, None, None
            import csv
            import glob
            for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mapping_pmc_per_kernel.csv")), reverse=True):
                for row in csv.DictReader(open(ppath)):
                    if pmc_key in row["kernel"] and row["counter"] == "TCC_ATOMIC_sum" and float(row["mean_per_dispatch"]) > 0:
                        reqs, req_src = float(row["mean_per_dispatch"]), os.path.relpath(ppath, ROOT)
                if reqs:
                    # counters of another shape / tiling beside this run's launch time would be a silently wrong fraction
                    meta = ppath.replace("_pmc_per_kernel.csv", "_pmc_meta.json")
                    req_ok = ("unverified (profile older than the signature file)" if not os.path.exists(meta) else
                              "matches this run" if json.load(open(meta)) == mapping_signature(rays, frames, S) else "STALE")
                    if req_ok == "STALE":
                        reqs = None
                    break
            if reqs and "sdfnet" in name:      # the counter's per-dispatch mean covers this kernel's two launches per iteration
                reqs = reqs * 2 * P / (P + 22 * rays)      # (composite points and eikonal points): the composite launch's share
            launches = sum(1 for n_, _, _, _ in prof if n_ == name) // PROF_ITERS
            per_launch_s = tms * 1e-3 / max(launches, 1)
            ATOMIC_CEILING = 20e9
            atomic = {"table_rows_per_launch": rows // max(launches, 1), "atomic_requests_per_launch": reqs,
                      "requests_source": req_src, "requests_profile": req_ok, "rows_per_request": round(rows / max(launches, 1) / reqs, 2) if reqs else None,
                      "requests_per_s": round(reqs / per_launch_s, 0) if reqs else None,
                      "ceiling_requests_per_s": ATOMIC_CEILING,
                      "frac_of_ceiling": round(reqs / per_launch_s / ATOMIC_CEILING, 3) if reqs else None,
                      "measured_ceiling": "fp32 atomics retire at ~20 G requests/s chip-wide on MI355X, independent of table size "
                                          "and contention (tools/micro/atomic_bench.hip, DESIGN.md 4b)"}
            roof = {"kernel": name, "bound": "atomic requests (memory-side RMW)", "achieved": atomic["requests_per_s"],
                    "peak": ATOMIC_CEILING, "unit": "requests/s", "frac": atomic["frac_of_ceiling"],
                    "traffic": None, "launches_per_iteration": launches, "avg_launch_us": round(per_launch_s * 1e6, 1),
                    "algorithmic_bytes_per_iteration": nbytes, "algorithmic_GBps": round(ach, 1),
                    "algorithmic_frac_of_hbm": round(ach / HBM_PEAK_GBS, 4), "atomic_scatter": atomic}
    return {"ms": round(dt * 1e3, 2), "rays": rays, "keyframes": frames, "samples_per_ray": S,
            "device_time_in_library": mapping_library_share(),
            "objective": "SLAMLoss with the weights of code/confs/replica/runconf_replica_1.conf (rgb, eikonal, smooth, ssi depth, "
                         "normals, patch warp [patch 1], flow over %d edges); stage fine / highfreq" % len(pairs),
            "eikonal_points": 22 * rays, "rays_per_s": round(rays / dt, 1), "engine": model.last_engine,
            "optimizer": "nicer_slam_amd.optim.Adam (1.1 GiB of parameters, dense)", "iters": iters,
            "final_loss": round(float(last), 6), "kernels_ms": {k: round(v, 3) for k, v in sorted(agg.items())},
            "roofline": roof, "cpu_baseline": cpu_mapping_baseline(model) if cpu else None}


def mapping_library_share():
    """Share of a mapping iteration's device time spent in this library's kernels (names in namespace nsa::) and launches per iteration,
    quoted from the newest committed rocprofv3 kernel summary of this leg (tools/profile_mapping.sh) -- a profile
    cannot be taken from inside the run, so the file is named."""
    import csv
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_mapping_kernel_stats.csv")), reverse=True)
    if not paths:
        return None
    # quoted only when the profile of that round was taken with this run's batch shape and tilings (the signature file the same script
    # writes next to it, tools/profile_mapping.sh): a stale profile is named, not quoted
    meta = paths[0].replace("_mapping_kernel_stats.csv", "_mapping_pmc_meta.json")
    try:
        fresh = json.load(open(meta)) == mapping_signature()
    except (OSError, ValueError):
        fresh = False
    if not fresh:
        return {"share": None, "launches_per_iteration": None, "source": os.path.relpath(paths[0], ROOT),
                "note": "STALE: taken with another batch shape / tiling set than this run's"}
    rows = list(csv.DictReader(open(paths[0])))
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    ours = sum(int(r["TotalDurationNs"]) for r in rows if "nsa::" in r["Name"])
    iters = max([int(r["Calls"]) for r in rows if "k_colour_bwd<true>" in r["Name"]] or [10])     # one launch of it per iteration
    return {"share": round(ours / total, 4), "launches_per_iteration": round(sum(int(r["Calls"]) for r in rows) / iters, 1),
            "source": os.path.relpath(paths[0], ROOT)}


def cpu_mapping_baseline(model, n=256, frames=8):
    """The oracle ("port") on the same mapping iteration -- forward, rgb L1 + 0.1 eikonal, backward to every trainable
    parameter (three tables, coarse SDF MLP, colour MLP) -- on a bounded sample of n rays over `frames` keyframes."""
    from oracle import render_ref as R
    torch.set_num_threads(min(32, os.cpu_count()))
    mk = R.make_grid_spec
    cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                         colour_grid=mk(16, 2, 16, 2048, 24), n_samples=64, n_samples_eval=640, n_samples_extra=32)
    frozen = "implicit_network.fine.lin"
    params = {}
    for k, v in model.state_dict().items():
        v = v.detach().cpu().clone()
        if v.is_floating_point() and k != "voxels" and not k.startswith(frozen) and "offsets" not in k:
            v.requires_grad_(True)
        params[k] = v
    g = torch.Generator().manual_seed(7)
    H, W = DS.img_res
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    K = K[None].repeat(frames, 1, 1)
    cams = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2]).repeat(frames, 1) + 0.01 * torch.randn(frames, 7, generator=g)
    pose = torch.stack([R.camera_from_tensor(c) for c in cams])
    times = []
    for it in range(4):
        idx = torch.randint(H * W, (frames, n // frames), generator=g)
        uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
        gt = torch.rand(n, 3, generator=g)
        draws = {"t_rand": torch.rand(n, 640, generator=g), "extra_idx": torch.randperm(640, generator=g)[:32],
                 "eik_idx": torch.randint(98, (n,), generator=g),
                 "eik_uniform": (torch.rand(10 * n, 3, generator=g) * 2 - 1) * cfg.scene_bounding_sphere,
                 "eik_jitter": torch.rand(11 * n, 3, generator=g)}
        for v in params.values():
            v.grad = None
        t0 = time.perf_counter()
        out = R.render(params, cfg, uv, pose, K, torch.zeros(64, 64, 64), draws, mode="mapping", training=True)
        loss = R.rgb_l1(out, gt) + 0.1 * ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean()
        loss.backward()
        times.append(time.perf_counter() - t0)
    timed = sorted(times[1:])
    med = timed[len(timed) // 2]
    return {"value": round(n / med, 1), "unit": "rays/s", "cores": max(torch.get_num_threads(), min(16, os.cpu_count() or 1)),
            "kind": "port", "sample": f"{n} rays over {frames} keyframes x (640 sampler + 98 composite) samples + {22 * n} eikonal "
                                      f"points, objective rgb L1 + 0.1 eikonal ONLY (the GPU leg beside it also runs the smooth, ssi-depth, "
                                      f"normal, patch-warp and flow terms and the Adam step: the CPU side does LESS work per ray), "
                                      f"fwd+bwd to every trainable parameter (no optimizer step), median of "
                                      f"{len(timed)} iterations after 1 warm-up ({round(sum(times), 1)} s of host time in all)"}


def cpu_baseline(args, model, conf):
    """The oracle (pure-PyTorch CPU restatement + C hash kernels = "port") timed on this box's host cores on a
    bounded sample of the same workload: `--cpu-rays` rays x (640 + S) samples, fwd+bwd to the pose gradient."""
    from oracle import render_ref as R
    n = args.cpu_rays
    # torch intra-op pool capped at 32: beyond that the small [n*768, 64] GEMMs of this sample only get slower
    # (256 threads measured 1 ray/s on the GPU box); the C hash kernels use min(16, cores) OpenMP threads.
    torch.set_num_threads(min(32, os.cpu_count()))
    mk = R.make_grid_spec
    cfg = R.RenderConfig(coarse=R.SdfNetSpec(mk(4, 8, 32, 32, 19), 2), fine=R.SdfNetSpec(mk(8, 4, 32, 128, 19), 4),
                         colour_grid=mk(16, 2, 16, 2048, 24), n_samples=args.samples - 34, n_samples_eval=640,
                         n_samples_extra=32)
    params = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    H, W = DS.img_res
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    vox = torch.zeros(64, 64, 64)
    times = []
    for it in range(7):
        idx = torch.randint(H * W, (1, n), generator=g)
        uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
        gt = torch.rand(n, 3, generator=g)
        draws = {"t_rand": torch.rand(n, 640, generator=g), "extra_idx": torch.randperm(640, generator=g)[:32],
                 "eik_idx": torch.randint(args.samples, (n,), generator=g)}
        cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2]).requires_grad_(True)
        t0 = time.perf_counter()
        out = R.render(params, cfg, uv, R.camera_from_tensor(cam).unsqueeze(0), K[None], vox, draws,
                       mode="tracking", training=True)
        R.rgb_l1(out, gt).backward()
        times.append(time.perf_counter() - t0)
    timed = sorted(times[2:])
    med = timed[len(timed) // 2]
    from oracle import hashenc
    omp = min(16, os.cpu_count() or 1)           # oracle/hashenc.py: nso_set_threads(min(16, cores))
    # BASELINE.md section 3 asks for torch.set_num_threads(os.cpu_count()): that figure as well, on a smaller sample of the same workload
    # (one timed iteration after one warm-up -- with every core of this box type the intra-op pool thrashes on these small GEMMs)
    all_cores = None
    if args.cpu_all_cores_rays <= 0:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_cpu_all_cores.json")), reverse=True):
            try:
                rec = json.load(open(path))
                all_cores = dict(rec["all_host_cores"], quoted_from=os.path.relpath(path, ROOT), measured_on_host_cores=rec.get("host_cores"),
                                 note="not re-measured in this run: one iteration costs ~48 s with every core of this box type")
                break
            except (OSError, ValueError, KeyError):
                continue
    if args.cpu_all_cores_rays > 0 and (os.cpu_count() or 1) > torch.get_num_threads():
        try:
            capped = torch.get_num_threads()
            torch.set_num_threads(os.cpu_count())
            m = args.cpu_all_cores_rays
            ts = []
            for it in range(2):
                idx = torch.randint(H * W, (1, m), generator=g)
                uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
                gt = torch.rand(m, 3, generator=g)
                draws = {"t_rand": torch.rand(m, 640, generator=g), "extra_idx": torch.randperm(640, generator=g)[:32],
                         "eik_idx": torch.randint(args.samples, (m,), generator=g)}
                cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2]).requires_grad_(True)
                t0 = time.perf_counter()
                out = R.render(params, cfg, uv, R.camera_from_tensor(cam).unsqueeze(0), K[None], vox, draws, mode="tracking", training=True)
                R.rgb_l1(out, gt).backward()
                ts.append(time.perf_counter() - t0)
            all_cores = {"value": round(m / ts[-1], 1), "unit": "rays/s", "threads": os.cpu_count(),
                         "sample": f"{m} rays, one iteration after one warm-up ({round(sum(ts), 1)} s of host time)"}
            torch.set_num_threads(capped)
        except Exception as e:      # a context figure must never take the line down
            all_cores = {"error": f"{type(e).__name__}: {e}"[:200]}
    return {"value": round(n / med, 1), "unit": "rays/s", "cores": max(torch.get_num_threads(), omp),
            "all_host_cores": all_cores,
            "threads": {"torch_intraop": torch.get_num_threads(), "c_hash_kernels_openmp": omp}, "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": f"{n} rays x (640 sampler + {args.samples} composite) samples, fwd+bwd to pose grad, "
                      f"median of {len(timed)} iterations after 2 warm-ups ({round(sum(times), 1)} s of host time in all); "
                      f"{torch.get_num_threads()} torch + {omp} OpenMP threads of {os.cpu_count()} host cores: more threads only "
                      f"slow this sample's [n x 768, 64] GEMMs down (all {os.cpu_count()} cores measured ~1 ray/s on this box type)"}


GATHER_CEILING_REQ_S = 55e9     # tools/micro/gather_bench.hip on MI355X (profiles/r04_gather_bench.txt): random 8-16 B row loads retire at
                                # ~55 G requests/s chip-wide, whatever the table size (128 MiB .. 768 MiB), depth or width


def colour_gather_roofline(agg, args):
    """Second roofline of the line: k_colour_fwd, the one HBM-resident gather (1 GiB colour table, 16 levels x 8 corner rows of 8 B
    per point), in the unit that bounds it -- memory-side REQUESTS per second (a request = one 64-byte line touched by one
    instruction) against the measured random-gather ceiling.  Time: this run's events.  Requests: TCC_HIT + TCC_MISS per launch from
    the newest committed `rocprofv3 --pmc` summary of this command (counters cannot be read inside the run; the source is named)."""
    if "k_colour_fwd" not in agg or args.rays != 1024 or args.samples != 128:
        return None
    import csv
    import glob
    tms, nbytes, n = agg["k_colour_fwd"]
    us = tms / n * 1e3
    hit = miss = None
    src = None
    for ppath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_per_kernel.csv")), reverse=True):
        vals = {}
        for row in csv.DictReader(open(ppath)):
            if "k_colour_fwd" in row["kernel"] and "bf16" not in row["kernel"] and row["counter"] in ("TCC_HIT", "TCC_MISS", "TCC_HIT_sum", "TCC_MISS_sum"):
                vals[row["counter"].replace("_sum", "")] = float(row["mean_per_dispatch"])
        if "TCC_HIT" in vals and "TCC_MISS" in vals:
            hit, miss, src = vals["TCC_HIT"], vals["TCC_MISS"], os.path.relpath(ppath, ROOT)
            break
    pts = args.rays * args.samples
    algo_rows = pts * 16 * 8
    out = {"kernel": "k_colour_fwd", "bound": "memory-side requests (random 8-byte rows)", "avg_launch_us": round(us, 2),
           "algorithmic_rows_per_launch": algo_rows, "algorithmic_bytes_per_launch": nbytes // n,
           "algorithmic_GBps": round(nbytes / n / (us * 1e-6) / 1e9, 1), "frac_of_hbm_peak": round(nbytes / n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "peak": GATHER_CEILING_REQ_S, "unit": "requests/s", "requests_source": src,
           "ceiling_source": "tools/micro/gather_bench.hip, profiles/r04_gather_bench.txt"}
    if hit is not None:
        req = hit + miss
        # The ceiling (tools/micro/gather_bench.hip) is for requests that ALL miss the L2; about half of this kernel's requests hit.
        # What the ceiling bounds is therefore the MISS rate: achieved = L2 misses per second (round 4 priced all requests against
        # it and printed a fraction of 1.23, which bounds nothing)
        out.update({"l2_requests_per_launch": req, "l2_hit_rate": round(hit / req, 3), "rows_per_request": round(algo_rows / req, 2),
                    "l2_requests_per_s": round(req / (us * 1e-6), 0),
                    "achieved": round(miss / (us * 1e-6), 0), "frac": round(miss / (us * 1e-6) / GATHER_CEILING_REQ_S, 3),
                    "note": "achieved = L2 MISSES per second against the all-miss random-gather ceiling (hits are served by the L2 "
                            "and are not what the ceiling measures)"})
    return out


def precision_modes_leg(args, device, K, steps=100):
    """The optional bf16-operand modes that BASELINE configs[2] / [4] name, at those configs' PER-GPU shapes (4096 x 128 over 8 GPUs =
    512 rays per GPU; 8192 x 192 over 8 GPUs = 1024 rays per GPU) and at the headline shape: ms per tracking iteration
    (KernelTracker, hipGraph) and the dominant kernel priced against the dense bf16 matrix line (2.5 PFLOP/s).  Context rows: the
    reference has no reduced-precision mode (fp32 only) and `value` stays the fp32 iteration.  Parity of these modes:
    tests/test_precision_oracle_gpu.py (bf16-emulating oracle)."""
    from nicer_slam_amd.hashencoder import backend as be
    from nicer_slam_amd.tracking import KernelTracker
    out = {"what": "optional reduced-precision MLP modes (BASELINE configs[2] / [4] per-GPU shapes); never the headline",
           "peak_tflops": MFMA_BF16_PEAK_TFLOPS}
    rows = (("bf16", 1024, 128, "headline shape, bf16 MLP operands"), ("bf16", 512, 128, "configs[2] per GPU (4096 x 128 / 8)"),
            ("bf16_colour", 1024, 192, "configs[4] per GPU (8192 x 192 / 8)"), ("fp32", 512, 128, "fp32 at the configs[2] per-GPU shape"),
            ("fp32", 1024, 192, "fp32 at the configs[4] per-GPU shape"))
    models = {}
    for prec, rays, samples, what in rows:
        key = f"{prec}_{rays}x{samples}"
        try:
            if samples not in models:
                a = argparse.Namespace(**vars(args))
                a.samples, a.precision, a.param_grads, a.engine = samples, "fp32", False, "auto"
                models[samples] = make_model(a, device)[0]
            model = models[samples]
            model.mlp_precision = prec
            model.__dict__.pop("_fused_pack", None)
            gen = torch.Generator(device=device).manual_seed(77)
            batches = [synth_batch(gen, rays, device) for _ in range(32)]
            cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device)
            tr = KernelTracker(model, K, rays, cam, lr=0.005, use_graph=True)
            with quiet_gc():
                for i in range(30):
                    tr.step(*batches[i % 32])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(steps):
                    tr.step(*batches[i % 32])
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / steps * 1e3
            eager = KernelTracker(model, K, rays, cam, lr=0.005, use_graph=False)
            for i in range(3):
                eager.step(*batches[i])
            be.PROFILE = []
            for i in range(10):
                eager.step(*batches[i])
            torch.cuda.synchronize()
            prof, be.PROFILE = be.PROFILE, None
            agg = {}
            for name, nbytes, e0, e1 in prof:
                v = agg.setdefault(name, [0.0, 0])
                v[0] += e0.elapsed_time(e1)
                v[1] += 1
            us = {k: v[0] / v[1] * 1e3 for k, v in agg.items()}
            row = {"what": what, "ms_per_step": round(ms, 4), "rays_per_s": round(rays / (ms * 1e-3), 1),
                   "kernels_us": {k: round(v, 1) for k, v in sorted(us.items())}}
            mlp = {k: v for k, v in us.items() if k in ALGO_MAC}
            if mlp:
                name = max(mlp, key=mlp.get)
                pts = rays * (640 if name == "k_sampler_sdf" else samples)
                bf16_kernel = prec == "bf16" or (prec == "bf16_colour" and "colour" in name)
                peak = MFMA_BF16_PEAK_TFLOPS if bf16_kernel else MFMA_F32_PEAK_TFLOPS
                ach = 2.0 * ALGO_MAC[name] * pts / (mlp[name] * 1e-6) / 1e12
                row["dominant"] = {"kernel": name, "avg_launch_us": round(mlp[name], 1), "achieved_tflops": round(ach, 1),
                                   "peak_tflops": peak, "frac": round(ach / peak, 4)}
            out[key] = row
            del tr, eager
        except Exception as e:      # a context leg must never take the headline down
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    models.clear()
    torch.cuda.empty_cache()
    return out


def small_shapes_leg(args, device, K, ms_1024, steps=100):
    """The strong-scaling tail on one GPU: an N-GPU strong-scaling run of the 1024-ray metric gives every rank
    1024 / N rays, and part of an iteration does not shrink with the ray count (per-ray scans that are latency bound, the last-workgroup
    tail, launch gaps, MLP kernels that no longer fill 256 CUs).  Rows: fp32 at 512 / 256 / 128 rays x 128 samples per GPU, (a) as the
    1-GPU tracker runs them (everything in one hipGraph) and (b) in the MULTI-RANK form of the step -- weighted 9-float message, graph
    without the Adam step, then all-reduce + nsa_adam_step_scaled -- on a ONE-rank RCCL group: per-rank compute + launch path of an N-GPU
    run, with the link latency of a real 2..8-rank all-reduce of 36 bytes NOT in it (no multi-GPU box).  `fit`: least squares
    ms = fixed + per_ray * R over R = 128..1024 -- `fixed` is the ray-count-independent part.  `predicted_strong_scaling`: the N-GPU
    speed-up of the 1024-ray metric these per-rank times allow (an upper bound: add the real exchange latency)."""
    import torch.distributed as dist
    from nicer_slam_amd.hashencoder import backend as be
    from nicer_slam_amd.tracking import KernelTracker
    out = {"what": "fp32 tracking iteration at the per-GPU shapes of an N-GPU strong-scaling run of the 1024-ray metric (one GPU)"}
    a = argparse.Namespace(**vars(args))
    a.samples, a.precision, a.param_grads, a.engine = 128, "fp32", False, "auto"
    model = make_model(a, device)[0]
    own_group = False
    try:
        if not dist.is_initialized():
            import socket
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
            sock.close()
            # an explicit store: under torchrun the environment tells init_process_group that the agent hosts the store
            # (TORCHELASTIC_USE_AGENT_STORE), and a tcp:// rendezvous of this process with itself then waits for its timeout
            store = dist.TCPStore("127.0.0.1", port, 1, is_master=True, use_libuv=False)
            dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=device)
            own_group = True
    except Exception as e:
        out["one_rank_group_error"] = f"{type(e).__name__}: {e}"[:200]

    def timed(rays, world):
        gen = torch.Generator(device=device).manual_seed(78)
        batches = [synth_batch(gen, rays, device) for _ in range(32)]
        cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device)
        tr = KernelTracker(model, K, rays, cam, lr=0.005, use_graph=True, world=world)
        with quiet_gc():
            for i in range(30):
                tr.step(*batches[i % 32])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                tr.step(*batches[i % 32])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        us = None
        if world == 1:
            eager = KernelTracker(model, K, rays, cam, lr=0.005, use_graph=False)
            for i in range(3):
                eager.step(*batches[i])
            be.PROFILE = []
            for i in range(10):
                eager.step(*batches[i])
            torch.cuda.synchronize()
            prof, be.PROFILE = be.PROFILE, None
            agg = {}
            for name, nbytes, e0, e1 in prof:
                v = agg.setdefault(name, [0.0, 0])
                v[0] += e0.elapsed_time(e1)
                v[1] += 1
            us = {k: round(v[0] / v[1] * 1e3, 1) for k, v in sorted(agg.items())}
            del eager
        ex = None
        if world > 1:        # the step's only exchange alone, back to back (host-issued: all-reduce + Adam launch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                tr._exchange()
            torch.cuda.synchronize()
            ex = round((time.perf_counter() - t0) / 50 * 1e6, 1)
        del tr
        return ms, us, ex

    pts = [(1024, ms_1024)]
    for rays in (512, 256, 128):
        key = f"fp32_{rays}x128"
        try:
            ms, us, _ = timed(rays, 1)
            row = {"ms_per_step": round(ms, 4), "rays_per_s": round(rays / (ms * 1e-3), 1), "kernels_us": us,
                   "kernels_sum_us": round(sum(us.values()), 1)}
            pts.append((rays, ms))
            if dist.is_initialized() and dist.get_world_size() == 1:
                ms_r, _, ex = timed(rays, 2)          # world = 2 selects the message form; the group has one rank
                row["multi_rank_form_ms_per_step"] = round(ms_r, 4)
                row["one_rank_exchange_us"] = ex
            out[key] = row
        except Exception as e:      # a context leg must never take the headline down
            out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if len(pts) >= 3:
        import numpy as np
        R = np.array([p[0] for p in pts], dtype=np.float64)
        T = np.array([p[1] for p in pts], dtype=np.float64)
        A = np.stack([np.ones_like(R), R], 1)
        (fixed, per_ray), *_ = np.linalg.lstsq(A, T, rcond=None)
        out["fit"] = {"fixed_us": round(float(fixed) * 1e3, 1), "us_per_ray": round(float(per_ray) * 1e3, 4),
                      "points_rays_ms": [[int(r), round(float(t), 4)] for r, t in pts],
                      "max_residual_us": round(float(np.abs(A @ np.array([fixed, per_ray]) - T).max()) * 1e3, 1)}
        pred = {}
        for n, rays in ((2, 512), (4, 256), (8, 128)):
            row = out.get(f"fp32_{rays}x128", {})
            t = row.get("multi_rank_form_ms_per_step") or row.get("ms_per_step")
            if t:
                pred[f"{n}_gpus"] = {"rays_per_gpu": rays, "per_rank_ms": t, "speedup_over_1_gpu": round(ms_1024 / t, 2),
                                     "efficiency": round(ms_1024 / t / n, 3)}
        out["predicted_strong_scaling"] = dict(pred, note="1024-ray metric; per-rank time measured on one GPU in the multi-rank form "
                                               "(1-rank RCCL group); the real 36-byte all-reduce latency over xGMI comes on top")
    if own_group:
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    del model
    torch.cuda.empty_cache()
    return out


def reference_shaped_gpu_leg(args, device, K, batches, value, steps=10):
    """Same-box GPU stand-in for north_star's 'reference single-GPU PyTorch path': the COMPOSED engine -- torch autograd around the
    stand-alone hash-encoder operator (C ABI section 1), i.e. the reference's op structure (code/model/network.py:78-347) launched
    eagerly, every parameter gradient computed as volsdf_train.py:417-427 does -- on the same 1024 x 128 batches.  It is a LOWER
    bound on the reference's own time: its hash operator is already this library's (the reference's CUDA extension cannot run
    here), and the loss / pose parametrisation use this library's kernels."""
    from nicer_slam_amd.tracking import TrackingStepper
    a = argparse.Namespace(**vars(args))
    a.param_grads, a.engine = True, "composed"
    try:
        model, _ = make_model(a, device)
        model.tracking_param_grads = True
        cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device)
        st = TrackingStepper(model, K, args.rays, cam, lr=0.005, use_graph=False, world=1)
        n = min(steps, len(batches))
        with quiet_gc():
            for i in range(3):
                st.step(*batches[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                st.step(*batches[i])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        out = {"ms_per_step": round(dt * 1e3, 3), "rays_per_s": round(args.rays / dt, 1), "engine": model.last_engine, "steps": n,
               "value_over_this": round(value / (args.rays / dt), 2),
               "what": "composed engine: torch autograd + the stand-alone HIP hash operator, eager, all parameter gradients "
                       "(reference op structure); a lower bound on the reference's time on this GPU"}
        del model, st
    except Exception as e:      # a context leg must never take the headline down
        out = {"error": f"{type(e).__name__}: {e}"[:300]}
    torch.cuda.empty_cache()
    return out


def dropin_leg(args, device, K, batches, steps=60):
    """The same tracking iteration driven the way the reference's loop drives it (volsdf_train.py:406-427): camera 7-vector
    -> get_camera_from_tensor -> SLAMNetwork.forward(mode="tracking") -> L1 -> loss.backward() -> torch.optim.Adam -- torch
    autograd around the fused autograd.Functions, no KernelTracker.  Two rows:
      faithful   every model parameter requires grad, exactly as volsdf_train.py builds the model (the reference computes and
                 discards all parameter gradients in tracking, :547 zeroes them before any use);
      pose_only  model.tracking_param_grads = False -- one attribute -- skips that discarded work; hipGraph-captured;
      pose_only_eager  the same launched eagerly, i.e. the reference's unmodified loop shape (it captures nothing): the ground-truth dict
                 goes to the model and to `tracking_loss` -- the SLAMLoss class resolved from the conf string `train.loss_class` with the
                 shipped `tracking_loss { ... }` block (volsdf_train.py:117-130, 415-424); the model's forward folds the L1 term in and
                 the loss class returns it (fused/track_graph.py, round 6);
      pose_only_eager_inline_l1  the same loop with the L1 term written out as torch ops on rgb_values (what rounds 3-5 timed as
                 pose_only_eager);
      pose_only_eager_hip_adam  pose_only_eager with ONE line changed: torch.optim.Adam -> nicer_slam_amd.optim.Adam (same semantics and
                 state_dict; one launch instead of torch's ~12 on the seven camera floats).
    Context numbers; `value` stays the KernelTracker iteration."""
    from nicer_slam_amd.tracking import TrackingStepper
    from nicer_slam_amd.utils.general import get_class
    out = {"driver": "SLAMNetwork.forward + loss_class + torch autograd + torch.optim.Adam (TrackingStepper)"}
    from nicer_slam_amd.optim import Adam as HipAdam
    # confs/replica/runconf_replica_1.conf:58-65 `tracking_loss { ... }`, class from `train.loss_class`
    tracking_loss_conf = dict(rgb_loss="torch.nn.L1Loss", eikonal_weight=0, smooth_weight=0, depth_weight=0, normal_l1_weight=0,
                              normal_cos_weight=0)
    for row, flag, graph, opt, seam in (("pose_only", False, True, None, False), ("pose_only_eager", False, False, None, True),
                                        ("pose_only_eager_inline_l1", False, False, None, False),
                                        ("pose_only_eager_hip_adam", False, False, HipAdam, True), ("faithful", True, False, None, False)):
        a = argparse.Namespace(**vars(args))
        a.param_grads = True                        # make_model leaves requires_grad as constructed
        model, _ = make_model(a, device)
        model.tracking_param_grads = flag
        cam = torch.tensor([1.0, 0, 0, 0, 0.1, 0.0, -0.2], device=device)
        try:
            loss_fn = get_class("nicer_slam_amd.model.loss.SLAMLoss")(model=model, **tracking_loss_conf) if seam else None
            st = TrackingStepper(model, K, args.rays, cam, lr=0.005, use_graph=graph, world=1, loss_fn=loss_fn,
                                 **({"opt_cls": opt} if opt else {}))
            n = min(steps, len(batches))
            with quiet_gc():
                for i in range(min(5, n)):
                    st.step(*batches[i])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    st.step(*batches[i])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
            out[row] = {"ms_per_step": round(dt * 1e3, 4), "rays_per_s": round(args.rays / dt, 1), "engine": model.last_engine,
                        "hip_graph": graph, "steps": n, "optimizer": "nicer_slam_amd.optim.Adam (one launch)" if opt else "torch.optim.Adam",
                        "objective": "loss_class = nicer_slam_amd.model.loss.SLAMLoss (tracking_loss block)" if seam else "inline torch L1"}
        except Exception as e:      # a context leg must never take the headline down
            out[row] = {"error": f"{type(e).__name__}: {e}"[:300]}
        del model
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    main()
