"""Size-independent properties at BASELINE's full size (1024 rays x 128 samples, E = 640, shipped grids incl. the
1 GiB colour table) where the CPU oracle is too slow to be the checker:
ray-permutation equivariance, shard invariance (the N > 1 decomposition), range / ordering invariants, visit-counter
conservation, and a finite-difference check of the pose gradient of the complete fused tracking iteration."""
import numpy as np
import pytest
import torch

from helpers import assert_close

pytestmark = pytest.mark.gpu


class _DS:
    img_res = (680, 1200)


@pytest.fixture(scope="module")
def world():
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    model = SLAMNetwork(replica_model_conf(94, 640, 32, use_warp_loss=False), dataset=_DS(), n_images=1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(3)
    with torch.no_grad():
        for enc, s in ((model.implicit_network.coarse.encoding, 0.02), (model.implicit_network.fine.encoding, 0.02),
                       (model.rendering_network.encoding, 0.3)):
            enc.embeddings.copy_((torch.rand(enc.embeddings.shape, device="cuda", generator=g) * 2 - 1) * s)
    for p in model.parameters():
        p.requires_grad_(False)
    model.engine = "fused"
    R = 1024
    idx = torch.randint(680 * 1200, (1, R), device="cuda", generator=g)
    uv = torch.stack([(idx % 1200).float(), (idx // 1200).float()], -1)
    K = torch.eye(4, device="cuda")
    K[0, 0] = K[1, 1] = 600.0
    K[0, 2], K[1, 2] = 599.5, 339.5
    draws = {"t_rand": torch.rand(R, 640, device="cuda", generator=g),
             "extra_idx": torch.randperm(640, device="cuda", generator=g)[:32],
             "eik_idx": torch.randint(128, (R,), device="cuda", generator=g)}
    return dict(model=model, uv=uv, K=K[None], draws=draws, R=R, gt=torch.rand(R, 3, device="cuda", generator=g))


def _render(w, cam, sel=None, mode="tracking", z_override=None):
    from nicer_slam_amd.utils.general import get_camera_from_tensor
    model = w["model"]
    sel = torch.arange(w["R"], device="cuda") if sel is None else sel
    d = w["draws"]
    model.draws = {"t_rand": d["t_rand"][sel], "extra_idx": d["extra_idx"], "eik_idx": d["eik_idx"][sel]}
    if z_override is not None:
        model.draws["z_vals_override"] = z_override
    out = model({"intrinsics": w["K"], "uv": w["uv"][:, sel], "pose": get_camera_from_tensor(cam).unsqueeze(0)},
                torch.zeros(1, dtype=torch.long, device="cuda"), {}, mode=mode, frame_idx=1)
    assert model.last_engine == "fused"
    return out


CAM = [1.0, 0.01, -0.02, 0.015, 0.1, 0.0, -0.2]


def test_invariants_permutation_and_shards(world):
    w = world
    cam = torch.tensor(CAM, device="cuda")
    with torch.no_grad():
        full = _render(w, cam)
        z, wts, rgb = full["z_vals"], full["weights"], full["rgb_values"].reshape(-1, 3)
        assert z.shape == (w["R"], 128) and bool((z[:, 1:] >= z[:, :-1]).all()) and bool((z >= 0).all())
        assert bool((wts >= 0).all()) and float(wts.sum(-1).max()) <= 1.0 + 1e-5
        assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 and float(full["rgb"].min()) > 0.0
        assert bool(torch.isfinite(full["depth_values"]).all()) and bool(torch.isfinite(full["normal_map"]).all())
        perm = torch.randperm(w["R"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        shuffled = _render(w, cam, perm)
        for k in ("rgb_values", "depth_values", "normal_map"):      # rays are independent: bit-exact equivariance
            assert torch.equal(shuffled[k].reshape(w["R"], -1), full[k].reshape(w["R"], -1)[perm]), k
        assert torch.equal(shuffled["z_vals"], z[perm])
        lo = _render(w, cam, torch.arange(0, 400, device="cuda"))    # two uneven shards = the N > 1 decomposition
        hi = _render(w, cam, torch.arange(400, w["R"], device="cuda"))
        for k in ("rgb_values", "depth_values", "normal_map"):
            both = torch.cat([lo[k].reshape(400, -1), hi[k].reshape(w["R"] - 400, -1)])
            assert torch.equal(both, full[k].reshape(w["R"], -1)), k


def test_visit_counter_conservation(world):
    w = world
    model = w["model"]
    before = model.voxels.clone()
    try:
        out = _render(w, torch.tensor(CAM, device="cuda"), mode="mapping")   # (eikonal samples need autograd: no no_grad)
        from nicer_slam_amd.utils import rend_util
        from nicer_slam_amd.utils.general import get_camera_from_tensor
        d, o = rend_util.get_camera_params(w["uv"], get_camera_from_tensor(torch.tensor(CAM, device="cuda")).unsqueeze(0), w["K"])
        x = o.reshape(1, 1, 3) + out["z_vals"].unsqueeze(-1) * d.reshape(-1, 1, 3)
        inside = int((~(x.abs() > 0.99).any(-1)).sum())
        added = float((model.voxels - before).sum())
        assert abs(added - inside) <= 2                               # samples exactly on the 0.99 test: ulp of the ray maths
        assert float((model.voxels - before).min()) >= 0.0
    finally:
        model.voxels = before


def test_pose_gradient_matches_central_differences(world):
    """d loss / d camera of the whole fused iteration vs central differences of the fused forward (fp32 kernels:
    step 2e-3 on a loss of ~0.25; agreement to a few per cent of the gradient norm is what fp32 differencing resolves)."""
    w = world
    cam = torch.tensor(CAM, device="cuda", requires_grad=True)
    out = _render(w, cam)
    zfix = out["z_vals"].detach()                                     # the sampler is not differentiated (reference: no_grad)
    loss = (out["rgb_values"].reshape(-1, 3) - w["gt"]).abs().mean() + 0.2 * out["depth_values"].mean()
    loss.backward()
    g = cam.grad.clone()
    rng = np.random.default_rng(0)
    for trial in range(3):
        v = torch.tensor(rng.standard_normal(7), device="cuda", dtype=torch.float32)
        v = v / v.norm()
        eps = 2e-3
        with torch.no_grad():
            vals = []
            for s in (+1, -1):
                o2 = _render(w, torch.tensor(CAM, device="cuda") + s * eps * v, z_override=zfix)
                vals.append(float((o2["rgb_values"].reshape(-1, 3) - w["gt"]).abs().mean() + 0.2 * o2["depth_values"].mean()))
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g * v).sum())
        assert abs(fd - an) <= 0.05 * float(g.norm()) + 1e-4, (trial, fd, an, float(g.norm()))


def test_tracking_recovers_a_perturbed_camera():
    """End to end through the hot path: target colours rendered by the model itself at a known camera, the camera perturbed by
    ~0.8 deg / 2.7 cm, 200 graph-replayed KernelTracker iterations with the reference's schedule (Adam + StepLR(50, 0.95),
    volsdf_train.py:393-446) on random pixel batches: the pose error must fall by more than 10x, and the arg-min-loss candidate
    must be as good (tools/demo_track.py)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import demo_track
    start, final, cand = demo_track.run(iters=200, verbose=False)
    assert final[0] < 0.1 * start[0] and final[1] < 0.1 * start[1], (start, final)
    assert cand[0] < 0.1 * start[0] and cand[1] < 0.1 * start[1], (start, cand)
