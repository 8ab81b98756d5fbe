"""Optional bf16-operand MLP modes of the fused engine (BASELINE configs[2] "bf16 MLP" and configs[4] "bf16 + fp32 SDF
head") against the default fp32-faithful mode, at those configs' per-GPU shapes.  Tolerance: rel 2e-2 on rgb
for the bf16 configurations, the fp32 SDF head held to the fp32 bound (here: bit-identical, the same kernels run)."""
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


class _DS:
    img_res = (680, 1200)


def _model(n_samples):
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(n_samples, 640, 32, use_warp_loss=False), dataset=_DS(), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc, s in ((model.implicit_network.coarse.encoding, 0.02), (model.implicit_network.fine.encoding, 0.02),
                       (model.rendering_network.encoding, 0.3)):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * s)
    for p in model.parameters():
        p.requires_grad_(False)
    model.engine = "fused"
    return model


def _run(model, R, S, precision, z_override=None):
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = torch.randint(680 * 1200, (1, R), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    gt = torch.rand(R, 3, device="cuda", generator=g)
    model.mlp_precision = precision
    model.draws = {"t_rand": torch.rand(R, 640, device="cuda", generator=g),
                   "extra_idx": torch.randperm(640, device="cuda", generator=g)[:32],
                   "eik_idx": torch.zeros(R, dtype=torch.long, device="cuda")}
    if z_override is not None:
        model.draws["z_vals_override"] = z_override
    cam = torch.tensor([1.0, 0.01, -0.02, 0.015, 0.1, 0.0, -0.2], device="cuda", requires_grad=True)
    out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam).unsqueeze(0)},
                torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode="tracking", frame_idx=1)
    assert model.last_engine == "fused" and out["z_vals"].shape == (R, S)
    (out["rgb_values"].reshape(-1, 3) - gt).abs().mean().backward()
    return out, cam.grad.clone()


@pytest.mark.parametrize("R,n_samples", [(512, 94), (1024, 158)])       # configs[2]: 4096x128 over 8 GPUs; configs[4]: 8192x192
def test_bf16_colour_mlp_keeps_the_fp32_sdf_head(R, n_samples):
    model = _model(n_samples)
    S = n_samples + 34
    ref, g_ref = _run(model, R, S, "fp32")
    out, g = _run(model, R, S, "bf16_colour")
    for k in ("z_vals", "sdf", "weights", "depth_values"):               # SDF networks untouched: the same kernels
        assert torch.equal(out[k], ref[k]), k
    assert_close(out["rgb_values"], ref["rgb_values"].detach().cpu().numpy(), 5e-3, 2e-2, "rgb_values (bf16 colour MLP)")
    assert float((out["rgb_values"] - ref["rgb_values"]).abs().max()) > 0            # the bf16 kernels really ran
    assert_close(g, g_ref.cpu().numpy(), 5e-2 * float(g_ref.abs().max()), 5e-2, "pose gradient")


@pytest.mark.parametrize("R,n_samples", [(512, 94)])
def test_bf16_everywhere(R, n_samples):
    model = _model(n_samples)
    S = n_samples + 34
    ref, g_ref = _run(model, R, S, "fp32")
    out, g = _run(model, R, S, "bf16", z_override=ref["z_vals"].detach())   # same sample positions: compare the networks
    # bf16 operands (8 significand bits) through the geometric-init SDF MLPs: |d sdf| of a few 1e-3 on values of O(0.1-1),
    # i.e. a surface shift of a few millimetres in scene units -- the reason configs[4] keeps the SDF head in fp32
    assert_close(out["sdf"], ref["sdf"].detach().cpu().numpy(), 2e-2, 2e-2, "sdf (bf16 SDF MLPs)")
    assert float((out["sdf"] - ref["sdf"]).abs().max()) > 1e-4
    assert_close(out["rgb_values"], ref["rgb_values"].detach().cpu().numpy(), 6e-2, 2e-2, "rgb_values")
    free, _ = _run(model, R, S, "bf16")                                  # the bf16 sampler itself: valid, sorted samples
    z = free["z_vals"]
    assert bool((z[:, 1:] >= z[:, :-1]).all()) and bool(torch.isfinite(z).all())
    with pytest.raises(ValueError):
        _run(model, 8, S, "fp16")


def test_bf16_colour_mapping_gradients_close_to_fp32():
    """The MAP backward kernels in the bf16-colour mode: every trainable gradient of a small mapping batch stays close
    (relative L2 error below 5 % per tensor) to the fp32-faithful run."""
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    res = {}
    for mode in ("fp32", "fp32", "bf16_colour"):     # first pass only fixes the sample positions: the two compared runs
                                                     # then take the same code path and consume the same random draws
        torch.manual_seed(4)
        model = SLAMNetwork(replica_model_conf(use_warp_loss=False), dataset=_DS(), n_images=1).cuda().freeze_fine_mlp()
        with torch.no_grad():
            for enc in (model.implicit_network.coarse.encoding, model.implicit_network.fine.encoding,
                        model.rendering_network.encoding):
                enc.embeddings.uniform_(-0.05, 0.05)
        model.train(True)
        model.engine = "fused"
        model.mlp_precision = mode
        if "z" in res:
            model.draws = {"z_vals_override": res["z"]}
        R = 128
        torch.manual_seed(8)
        idx = torch.randint(680 * 1200, (1, R), device="cuda")
        uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
        K = torch.eye(4, device="cuda")
        K[0, 0] = K[1, 1] = 600.0
        K[0, 2], K[1, 2] = 599.5, 339.5
        cam = torch.tensor([[1.0, 0.01, -0.02, 0.015, 0.1, 0.0, -0.2]], device="cuda")
        out = model({"intrinsics": K[None], "uv": uv, "pose": get_camera_from_tensor(cam)}, torch.zeros(1, dtype=torch.long, device="cuda"),
                    {}, mode="mapping", stage="fine", color_stage="highfreq", frame_idx=5)
        assert model.last_engine == "fused"
        res.setdefault("z", out["z_vals"].detach())
        loss = (out["rgb_values"] - 0.4).abs().mean() + 0.1 * ((out["grad_theta"].norm(2, dim=1) - 1) ** 2).mean()
        loss.backward()
        res[mode] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert set(res["fp32"]) == set(res["bf16_colour"]) and len(res["fp32"]) >= 12
    # judged per tensor in the L2 norm (bias / weight gradients are signed sums over ~1e4 points)
    errs = {n: float((res["bf16_colour"][n] - g).norm() / (g.norm() + 1e-12)) for n, g in res["fp32"].items()}
    for n, err in errs.items():
        assert err < 0.05, (n, err, errs)
    assert max(errs.values()) > 1e-5                  # the bf16 colour backward really ran


def test_bf16_resident_quad_kernels_are_bit_identical_to_the_staged_forms(tmp_path):
    """The bf16 build's paired SDF forward and fine backward run as resident-weight, barrier-free persistent kernels (mlp16.hpp::
    ResidentSeq); NSA_BF16_RESIDENT=0 selects the staged forms, which include the SAME per-tile statements (sdfnet4_*_body.inc): the
    forward tensors of a tracking iteration must agree bit for bit, the pose gradient to the last bits.  The switch is read once per process, hence two subprocesses."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_precision_gpu as t
model = t._model(94)
out, g = t._run(model, 300, 128, "bf16")
torch.save({"g": g.cpu(), **{k: out[k].detach().cpu() for k in ("sdf", "rgb", "weights", "rgb_values", "depth_values", "normal_map", "z_vals")}}, sys.argv[1])
''' % (root, os.path.join(root, "tests"))
    res = {}
    for flag in ("1", "0"):
        path = str(tmp_path / f"res{flag}.pt")
        env = dict(os.environ, NSA_BF16_RESIDENT=flag)
        p = subprocess.run([sys.executable, "-c", script, path], capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        res[flag] = torch.load(path)
    diffs = {k: float((v - res["0"][k]).abs().max()) for k, v in res["1"].items()}
    print("resident vs staged, max |difference| per tensor:", diffs)
    for k in ("sdf", "rgb", "weights", "rgb_values", "depth_values", "normal_map", "z_vals"):     # the forward: bit for bit
        assert diffs[k] == 0.0, diffs
    # the pose gradient passes through the fine backward, a DIFFERENT kernel function in the two forms: the same statements, but the
    # compiler is free to contract a * b + c differently in each -- last-bit differences are allowed, nothing more
    assert diffs["g"] <= 1e-6 * float(res["0"]["g"].abs().max()), diffs
    assert float(res["1"]["g"].abs().max()) > 0
