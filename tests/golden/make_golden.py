#!/usr/bin/env python
"""Generate the committed golden vectors by running the REFERENCE's own Python on CPU.

Runs only in the build container (needs /root/reference).  The reference's native CUDA
extension cannot execute here, so ``hashencoder.backend._backend`` is the CPU oracle
(oracle/hashenc_oracle.c); everything above it -- hashgrid.py's autograd Functions and
HashEncoder, model/{network,base_networks,ray_sampler,density,embedder}.py,
utils/{rend_util,general}.py -- is the reference's unmodified code.  The hash/index rule
itself is pinned independently by known-answer vectors (tests/test_oracle_kat.py) and by the
reference's pure-torch twin ``HashEncoder.torch_forward`` (case ``twin_*`` below).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Fixtures are DATA: inputs (incl. every parameter tensor and every captured RNG draw) and the
reference's outputs/gradients.  No reference source is stored.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shims  # noqa: E402
from oracle import hashenc  # noqa: E402

hg = ref_shims.install(hashenc.OracleBackend())
torch.Tensor.get_device = lambda self: "cpu"  # general.quad2rotation does .to(t.get_device())
ref_network = ref_shims.import_ref("model.network")
ref_general = ref_shims.import_ref("utils.general")
ref_rend = ref_shims.import_ref("utils.rend_util")


# ------------------------------------------------------------------ RNG capture
class DrawLog:
    def __init__(self):
        self.draws = []


@contextlib.contextmanager
def capture_draws(log):
    """Record every CPU random draw the reference makes inside forward (ray_sampler.py:58,148,158;
    network.py:318-330), in call order."""
    o_rand, o_perm, o_int, o_like, o_unif = (torch.rand, torch.randperm, torch.randint,
                                             torch.rand_like, torch.Tensor.uniform_)

    def rec(kind, t):
        log.draws.append((kind, t.detach().clone()))
        return t

    torch.rand = lambda *a, **k: rec("rand", o_rand(*a, **k))
    torch.randperm = lambda *a, **k: rec("randperm", o_perm(*a, **k))
    torch.randint = lambda *a, **k: rec("randint", o_int(*a, **k))
    torch.rand_like = lambda *a, **k: rec("rand_like", o_like(*a, **k))
    torch.Tensor.uniform_ = lambda self, *a, **k: rec("uniform_", o_unif(self, *a, **k))
    try:
        yield log
    finally:
        torch.rand, torch.randperm, torch.randint, torch.rand_like = o_rand, o_perm, o_int, o_like
        torch.Tensor.uniform_ = o_unif


# ------------------------------------------------------------------ configs
# the model subtree differs between the shipped conf families in exactly these keys (confs/replica/runconf_replica_1.conf:96,118
# vs confs/7scenes/runconf_7scenes_1.conf:98,111,114,122,136 = confs/azure/*.conf)
FAMILY = {"replica": dict(coarse=dict(bias=0.6), fine=dict(geometric_init=True)),
          "7scenes": dict(coarse=dict(bias=1.0, concat_coarse_feature=False),
                          fine=dict(geometric_init=False, clamp=False, concat_coarse_feature=False))}


def model_conf(coarse_grid, fine_grid, n_samples, n_eval, n_extra, warp=False, family="replica"):
    def sdf(dims, g, which):
        d = dict(d_in=3, d_out=1, dims=dims, geometric_init=True, bias=0.6, skip_in=[],
                 weight_norm=True, multires=6, inside_outside=True, use_grid_feature=True,
                 base_size=g[0], end_size=g[1], logmap=g[2], num_levels=g[3], level_dim=g[4],
                 divide_factor=1.0, embedding_method="nerf")
        d.update(FAMILY[family][which])
        return d
    return ref_shims.Conf(
        feature_vector_size=64, scene_bounding_sphere=1.0, use_warp_loss=bool(warp),
        mapping_patchsizes=[1, 5] if warp else [1], tracking_patchsizes=[1], sampling_method="important",
        density_method="volsdf_gridpredefined",
        implicit_network=dict(coarse=sdf([64], coarse_grid, "coarse"), fine=sdf([64, 64, 64], fine_grid, "fine")),
        rendering_network=dict(mode="idr", d_in=9, d_out=3, dims=[64, 64], weight_norm=True,
                               multires_view=4, per_image_code=False, use_grid_feature=True),
        gridpredefinedensity={},
        ray_sampler=dict(near=0.0, N_samples=n_samples, N_samples_eval=n_eval, N_samples_extra=n_extra),
    )


class _DS:
    img_res = (680, 1200)


class _DS7:                      # confs/7scenes/runconf_7scenes_1.conf:68-71
    img_res = (480, 640)


def build_model(seed, coarse_grid, fine_grid, colour_grid, n_samples, n_eval, n_extra, emb_scale, warp=False, ds=None,
                rand_v=0.0, family="replica"):
    """Reference SLAMNetwork with reduced-size tables.  The colour encoder is hard-coded to a
    1 GiB table (base_networks.py:265-284); it is swapped for the reference's own HashEncoder
    class with a small geometry that keeps 16 levels x 2 features."""
    torch.manual_seed(seed)
    conf = model_conf(coarse_grid, fine_grid, n_samples, n_eval, n_extra, warp, family)
    # build with a throw-away tiny colour grid to avoid allocating 1 GiB: patch the class default
    RN = ref_shims.import_ref("model.base_networks").RenderingNetwork
    HE = hg.HashEncoder
    orig = HE.__init__

    def small_init(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                   log2_hashmap_size=19, desired_resolution=None):
        if log2_hashmap_size == 24:  # the hard-coded colour grid
            base_resolution, desired_resolution, log2_hashmap_size = colour_grid
        orig(self, input_dim, num_levels, level_dim, per_level_scale, base_resolution, log2_hashmap_size,
             desired_resolution)

    HE.__init__ = small_init
    try:
        model = ref_network.SLAMNetwork(conf, dataset=ds or _DS(), n_images=4)
    finally:
        HE.__init__ = orig
    g = torch.Generator().manual_seed(seed + 100)
    for enc, sc in ((model.implicit_network.coarse.encoding, emb_scale[0]),
                    (model.implicit_network.fine.encoding, emb_scale[1]),
                    (model.rendering_network.encoding, emb_scale[2])):
        enc.embeddings.data = (torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * sc
    # perturb weight_g so weight-norm is exercised away from its init point
    for n, p in model.named_parameters():
        if n.endswith("weight_g"):
            p.data = p.data * (1 + 0.1 * (torch.rand(p.shape, generator=g) - 0.5))
    if rand_v:
        # The geometric initialisation zeroes every first-layer column except x,y,z (base_networks.py:127-146), so a freshly
        # built SDF network ignores its positional encoding and its grid features: their gradients -- table scatter, Jacobian
        # terms, the double backward through grad sdf -- are identically zero.  The "_rw" cases add noise to every weight_v
        # (as a trained / pretrained network has) so that those paths carry signal through the reference's autograd.
        for n, p in model.named_parameters():
            if n.startswith("implicit_network") and n.endswith("weight_v"):
                first = ".lin0." in n
                p.data = p.data + (rand_v if first else 0.2 * rand_v) * torch.randn(p.shape, generator=g)
    return model, conf


def synth_inputs(seed, bs, n_pix, res=_DS.img_res, cam_k=(600.0, 599.5, 339.5)):
    g = torch.Generator().manual_seed(seed)
    H, W = res
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = cam_k[0]
    K[0, 2], K[1, 2] = cam_k[1], cam_k[2]
    K[0, 1] = 0.3  # exercise the skew term of lift()
    idx = torch.randint(H * W, (bs, n_pix), generator=g)
    uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    cam = torch.zeros(bs, 7)
    cam[:, 0] = 1.0
    cam[:, :4] += 0.05 * torch.randn(bs, 4, generator=g)   # NOT unit: two_s = 2/|q|^2 matters
    cam[:, 4:] = torch.tensor([0.1, 0.0, -0.2]) + 0.05 * torch.randn(bs, 3, generator=g)
    return uv, cam, K[None].repeat(bs, 1, 1)


def t2n(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


def objective(out, gt, mode):
    """Scalar used to pull gradients through every differentiable output (terms follow
    model/loss.py:57-110 in form; the weights are arbitrary but fixed)."""
    loss = (out["rgb_values"].reshape(-1, 3) - gt["rgb"]).abs().mean()
    if mode == "mapping":
        loss = loss + 0.1 * (out["depth_values"].reshape(-1, 1) - gt["depth"]).abs().mean()
        n = torch.nn.functional.normalize(out["normal_map"].reshape(-1, 3), p=2, dim=-1)
        loss = loss + 0.05 * (n - gt["normal"]).abs().sum(-1).mean() + 0.05 * (1 - (n * gt["normal"]).sum(-1)).mean()
        g1, g2 = out["grad_theta"], out["grad_theta_nei"]
        loss = loss + 0.1 * ((g1.norm(2, dim=1) - 1) ** 2).mean()
        n1 = g1 / (g1.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        n2 = g2 / (g2.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        loss = loss + 0.005 * torch.norm(n1 - n2, dim=-1).mean()
        loss = loss + 0.01 * out["entropy"]
    return loss


def full_case(name, seed, mode, stage, color_stage, bs, n_pix, training=True, poisson=False,
              grids=None, samples=(10, 32, 6), rand_v=0.0, family="replica"):
    """family = "7scenes": the model subtree of the 7-Scenes / Azure confs (coarse sphere radius 1.0, fine SDF MLP at nn.Linear's
    default initialisation -- which, unlike the geometric one, feeds every positional-encoding and grid column from the start),
    480 x 640 frames and the Kinect camera (f = 585, principal point (320, 240))."""
    coarse_grid, fine_grid, colour_grid = grids or ((4, 4, 8, 4, 8), (4, 32, 10, 8, 4), (4, 64, 10))
    seven = family == "7scenes"
    model, conf = build_model(seed, coarse_grid, fine_grid, colour_grid, *samples,
                              emb_scale=(0.05, 0.05, 0.5) if not rand_v else (0.3, 0.3, 0.5), rand_v=rand_v, family=family,
                              ds=_DS7() if seven else None)
    if seven:
        # The far sample of every ray sits ON a cube face, where the grids' in-range test hangs on the last ulp of o + z d;
        # this family's fine network listens to its grid features from the start, so a ray whose far sample carries weight would make
        # the fixture's outputs and gradients depend on that ulp.  The coarse sdf bias is therefore moved 0.3 inwards from its initial
        # 1.0 (a map that has settled inside its initial radius): every ray meets the surface inside the cube and its far sample has
        # no weight.  (A parameter value, i.e. an input of the fixture -- the family's structure is untouched.)
        with torch.no_grad():
            model.implicit_network.coarse.lin1.bias[0] -= 0.3
    model.train(training)
    uv, cam, K = synth_inputs(seed + 1, bs, n_pix, *((_DS7.img_res, (585.0, 320.0, 240.0)) if seven else ()))
    g = torch.Generator().manual_seed(seed + 2)
    if poisson:
        model.voxels.copy_(torch.poisson(torch.full(model.voxels.shape, 50.0), generator=g))
    voxels_in = model.voxels.clone()
    cam = cam.clone().requires_grad_(True)
    pose = ref_general.get_camera_from_tensor(cam)
    gt = {"rgb": torch.rand(bs * n_pix, 3, generator=g), "depth": torch.rand(bs * n_pix, 1, generator=g) * 2,
          "normal": torch.nn.functional.normalize(torch.randn(bs * n_pix, 3, generator=g), dim=-1)}
    log = DrawLog()
    torch.manual_seed(seed + 3)
    with capture_draws(log):
        out = model({"intrinsics": K, "uv": uv, "pose": pose}, torch.arange(bs), {}, mode=mode,
                    stage=stage, color_stage=color_stage, frame_idx=1)
    rec = {"in_uv": uv, "in_cam": cam, "in_K": K, "in_pose": pose, "in_voxels": voxels_in,
           "meta_mode": mode, "meta_stage": stage, "meta_color_stage": color_stage,
           "meta_training": int(training), "meta_samples": np.array(samples),
           "meta_coarse_grid": np.array(coarse_grid), "meta_fine_grid": np.array(fine_grid),
           "meta_colour_grid": np.array(colour_grid), "meta_family": family,
           "meta_frame_res": np.array(_DS7.img_res if seven else _DS.img_res)}
    kinds = [k for k, _ in log.draws]
    if training:
        assert kinds[:3] == ["rand", "randperm", "randint"], kinds
        rec["draw_t_rand"], rec["draw_extra_idx"] = log.draws[0][1], log.draws[1][1][: samples[2]]
        rec["draw_eik_idx"] = log.draws[2][1]
        if mode == "mapping":
            assert kinds[3:] == ["uniform_", "rand_like"], kinds
            rec["draw_eik_uniform"], rec["draw_eik_jitter"] = log.draws[3][1], log.draws[4][1]
    else:
        assert kinds == ["randint"], kinds
        rec["draw_eik_idx"] = log.draws[0][1]
    for k, v in model.state_dict().items():
        rec["param_" + k] = v
    for k in ("rgb", "rgb_values", "depth_values", "z_vals", "depth_vals", "sdf", "weights", "entropy",
              "normal_map", "grad_theta", "grad_theta_nei"):
        if k in out:
            rec["out_" + k] = out[k]
    rec["out_voxels"] = model.voxels.clone()
    if training:
        loss = objective(out, gt, mode)
        loss.backward()
        rec["out_loss"] = loss
        rec["grad_cam"] = cam.grad
        for n, p in model.named_parameters():
            rec["grad_" + n] = p.grad if p.grad is not None else torch.zeros(0)
        for k, v in gt.items():
            rec["gt_" + k] = v
    # a1/a17 on their own
    d, o = ref_rend.get_camera_params(uv, pose.detach(), K)
    rec["out_ray_dirs"], rec["out_cam_loc"] = d, o
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, "loss" in locals() and float(loss), {k: tuple(v.shape) for k, v in rec.items()
                                                     if k.startswith("out_") and hasattr(v, "shape")})


def encoder_case(name, seed, L, C, base, end, logmap, n_pts, emb_scale=1.0):
    """Function-level vectors through the reference's wrappers (hashgrid.py:13-134,199-215):
    forward, Jacobian contraction, table scatter, and the two second-backward products."""
    torch.manual_seed(seed)
    enc = hg.HashEncoder(input_dim=3, num_levels=L, level_dim=C, per_level_scale=2, base_resolution=base,
                         log2_hashmap_size=logmap, desired_resolution=end)
    g = torch.Generator().manual_seed(seed)
    enc.embeddings.data = (torch.rand(enc.embeddings.shape, generator=g) * 2 - 1) * emb_scale
    x = torch.rand(n_pts, 3, generator=g) * 2 - 1
    x[0] = torch.tensor([-1.0, -1.0, -1.0])      # maps to 0 exactly
    x[1] = torch.tensor([1.0, 1.0, 1.0])         # maps to 1 exactly (corner +1 rows, weight 0)
    x[2] = torch.tensor([1.0, -0.3, 0.25])
    x[3] = torch.tensor([1.0001, 0.0, 0.0])      # out of range -> zeros
    x[4] = torch.tensor([0.2, -1.5, 0.0])        # out of range -> zeros
    x[5] = x[6].clone()                          # duplicate cell (scatter contention)
    x = x.requires_grad_(True)
    v = torch.randn(n_pts, L * C, generator=g).requires_grad_(True)   # upstream grad of the value
    q = torch.randn(n_pts, 3, generator=g)                            # upstream grad of grad_inputs
    r = torch.randn(n_pts, L * C, generator=g)
    y = enc(x)
    (gx,) = torch.autograd.grad(y, x, v, create_graph=True)
    first_emb = torch.autograd.grad(y, enc.embeddings, v, retain_graph=True)[0]
    loss2 = (gx * q).sum() + (y * r).sum()
    loss2.backward()
    rec = dict(meta_grid=np.array([L, C, base, end, logmap]), in_x=x, in_v=v, in_q=q, in_r=r,
               param_embeddings=enc.embeddings, param_offsets=enc.offsets,
               meta_per_level_scale=np.float64(enc.per_level_scale),
               out_y=y, out_gx=gx, out_first_emb=first_emb, out_emb_grad=enc.embeddings.grad,
               out_v_grad=v.grad, out_x_grad=x.grad)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, tuple(enc.embeddings.shape), enc.offsets.tolist())


def twin_case(name, seed, L, C, base, end, n_pts):
    """The reference's pure-torch twin (hashgrid.py:217-299): dense levels, interior points."""
    torch.manual_seed(seed)
    enc = hg.HashEncoder(input_dim=3, num_levels=L, level_dim=C, per_level_scale=2, base_resolution=base,
                         log2_hashmap_size=19, desired_resolution=end)
    g = torch.Generator().manual_seed(seed)
    enc.embeddings.data = torch.rand(enc.embeddings.shape, generator=g) * 2 - 1
    # the twin sizes each level from a float64 scale, the kernel from float32 exp2f; they only
    # describe the same grid when both give the same resolution -- true for this geometry
    # (and for the shipped 32->32, 32->128, 16->2048 grids), checked here.
    for lv in range(L):
        f64 = int(np.ceil(np.exp2(lv * np.log2(enc.per_level_scale)) * base - 1)) + 1
        assert f64 == hashenc.level_geometry(enc.offsets.numpy(), lv, np.log2(enc.per_level_scale), base)[2]
    x = (torch.rand(n_pts, 3, generator=g) * 2 - 1) * 0.98
    x.requires_grad_(True)
    y = enc.torch_forward(x)
    v = torch.randn(y.shape, generator=g).requires_grad_(True)
    q = torch.randn(n_pts, 3, generator=g)
    # stock autograd through the twin, twice: first order (J^T v, table scatter) and the two second-order products the
    # CUDA path keeps -- d/dv and d/dtable of s = <J^T v, q>  (hashencoder.cu:405-625; d s / d x is what hashgrid.py:134
    # drops, so it is NOT recorded).  Pins kernel_grid_backward and both second-backward kernels independently of the
    # C restatement.
    (gx,) = torch.autograd.grad(y, x, v, create_graph=True)
    (first_emb,) = torch.autograd.grad(y, enc.embeddings, v, retain_graph=True)
    s = (gx * q).sum()
    v_grad2, emb_grad2 = torch.autograd.grad(s, [v, enc.embeddings])
    rec = dict(meta_grid=np.array([L, C, base, end, 19]), in_x=x, in_v=v, in_q=q, param_embeddings=enc.embeddings,
               param_offsets=enc.offsets, meta_per_level_scale=np.float64(enc.per_level_scale),
               out_y=y, out_gx=gx, out_first_emb=first_emb, out_v_grad2=v_grad2, out_emb_grad2=emb_grad2)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, tuple(enc.embeddings.shape))


def sparse_colour_case(name, seed, n_pts):
    """Real colour-grid geometry (16x2, 16->2048, 2^24 rows/level, base_networks.py:265-284) incl.
    the uint32 stride wrap at resolution 2048.  The 1 GiB table is stored sparsely: only rows
    the points touch are kept (value = f(row)), everything else is zero on both sides."""
    L, C, base, end, logmap = 16, 2, 16, 2048, 24
    pls = np.exp2(np.log2(end / base) / (L - 1))
    offs, tot = [], 0
    for i in range(L):
        res = int(np.ceil(base * pls ** i))
        offs.append(tot)
        tot += min(2 ** logmap, res ** 3)
    offs.append(tot)
    offsets = torch.tensor(offs, dtype=torch.int32)
    g = torch.Generator().manual_seed(seed)
    x01 = torch.rand(n_pts, 3, generator=g)
    x01[0] = torch.tensor([1.0, 1.0, 1.0])
    x01[1] = torch.tensor([0.0, 0.5, 1.0])
    S = np.log2(pls)
    # enumerate touched rows with the oracle's own index rule, fill them from a row-keyed formula
    emb = torch.zeros(tot, C)   # lazily committed by the OS; only touched rows are written
    rows = set()
    for lv in range(L):
        row0, nrows, res, scale = hashenc.level_geometry(offs, lv, S, base)
        cell = np.floor(x01.numpy() * np.float32(scale)).astype(np.uint32)
        for p in range(n_pts):
            for corner in range(8):
                q = [int(cell[p, d]) + ((corner >> d) & 1) for d in range(3)]
                rows.add(row0 + hashenc.level_row(nrows, res, q))
    rows = torch.tensor(sorted(rows), dtype=torch.long)
    vals = torch.stack([torch.sin(rows.double() * 0.37), torch.cos(rows.double() * 0.11)], -1).float()
    emb[rows] = vals
    emb.requires_grad_(True)
    x01.requires_grad_(True)
    y = hg.hash_encode(x01, emb, offsets, pls, base, True)
    v = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x01, v)
    rec = dict(meta_grid=np.array([L, C, base, end, logmap]), meta_per_level_scale=np.float64(pls),
               in_x01=x01, in_v=v, param_rows=rows, param_vals=vals, param_offsets=offsets,
               out_y=y, out_gx=gx)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, len(rows), "rows touched of", tot)


class _SmallDS:
    img_res = (40, 60)


def warp_case(name, seed, bs=2, n_pix=6):
    """Mapping mode with the patch-warp block (network.py:167-279) on 40x60 images, patch sizes 1 and 5."""
    coarse_grid, fine_grid, colour_grid = (4, 4, 8, 4, 8), (4, 32, 10, 8, 4), (4, 64, 10)
    samples = (10, 32, 6)
    model, conf = build_model(seed, coarse_grid, fine_grid, colour_grid, *samples, emb_scale=(0.05, 0.05, 0.5),
                              warp=True, ds=_SmallDS())
    model.train(True)
    H, W = _SmallDS.img_res
    g = torch.Generator().manual_seed(seed + 1)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 30.0
    K[0, 2], K[1, 2] = 29.5, 19.5
    idx = torch.randint(H * W, (bs, n_pix), generator=g)
    uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    uv[0, 0] = torch.tensor([1.0, 1.0])          # patch partly outside the image
    cam = torch.zeros(bs, 7)
    cam[:, 0] = 1.0
    cam[:, :4] += 0.03 * torch.randn(bs, 4, generator=g)
    cam[:, 4:] = torch.tensor([0.1, 0.0, -0.2]) + 0.03 * torch.randn(bs, 3, generator=g)
    cam = cam.requires_grad_(True)
    pose = ref_general.get_camera_from_tensor(cam)
    Kb = K[None].repeat(bs, 1, 1)
    full_rgb = torch.rand(bs, H * W, 3, generator=g)
    full_depth = 1.0 + 0.02 * torch.rand(bs, H * W, 1, generator=g)
    full_depth[:, : H * W // 3] += torch.rand(bs, H * W // 3, 1, generator=g)    # a non-flat region
    gt_rgb = torch.rand(bs * n_pix, 3, generator=g)
    log = DrawLog()
    torch.manual_seed(seed + 3)
    with capture_draws(log):
        out = model({"intrinsics": Kb, "uv": uv, "pose": pose}, torch.arange(bs),
                    {"full_rgb": full_rgb, "full_depth": full_depth}, mode="mapping", stage="fine",
                    color_stage="highfreq", frame_idx=1)
    rec = {"in_uv": uv, "in_cam": cam, "in_K": Kb, "in_pose": pose, "in_voxels": torch.zeros(64, 64, 64),
           "in_full_rgb": full_rgb, "in_full_depth": full_depth, "gt_rgb": gt_rgb,
           "meta_mode": "mapping", "meta_stage": "fine", "meta_color_stage": "highfreq", "meta_training": 1,
           "meta_samples": np.array(samples), "meta_coarse_grid": np.array(coarse_grid),
           "meta_fine_grid": np.array(fine_grid), "meta_colour_grid": np.array(colour_grid),
           "meta_img_res": np.array([H, W])}
    rec["draw_t_rand"], rec["draw_extra_idx"] = log.draws[0][1], log.draws[1][1][: samples[2]]
    rec["draw_eik_idx"], rec["draw_eik_uniform"], rec["draw_eik_jitter"] = log.draws[2][1], log.draws[3][1], log.draws[4][1]
    for k, v in model.state_dict().items():
        rec["param_" + k] = v
    loss = (out["rgb_values"].reshape(-1, 3) - gt_rgb).abs().mean()
    for ps, (gt_w, samp, mask, ray_mask) in out["warp_output"].items():
        rec[f"out_warp{ps}_gt"], rec[f"out_warp{ps}_sampled"], rec[f"out_warp{ps}_mask"] = gt_w, samp, mask
        if ray_mask is not None:
            rec[f"out_warp{ps}_raymask"] = ray_mask
        loss = loss + 0.5 * ((gt_w - samp).abs().sum(-1) * mask.float()).sum() / (mask.float().sum() + 1)
    loss.backward()
    rec["out_loss"], rec["grad_cam"] = loss, cam.grad
    rec["out_z_vals"], rec["out_rgb_values"], rec["out_depth_values"] = out["z_vals"], out["rgb_values"], out["depth_values"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, float(loss), {k: tuple(v.shape) for k, v in rec.items() if k.startswith("out_warp")})


class _CutDepth(torch.overrides.TorchFunctionMode):
    """Replaces the result of `rendered_depth = depth_values.unsqueeze(2)` (network.py:149) by a fresh leaf holding the same
    values: the re-projection blocks downstream (flow :153-165, patch warp :167-279) are then a function of (depth leaf, pose)
    alone, so their gradients can be recorded separately from the renderer's."""

    def __init__(self, n_rays):
        super().__init__()
        self.n_rays, self.leaf = n_rays, None

    def __torch_function__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        if (self.leaf is None and func is torch.Tensor.unsqueeze and len(args) == 2 and args[1] == 2
                and tuple(args[0].shape) == (self.n_rays, 1) and args[0].requires_grad):
            self.leaf = out.detach().clone().requires_grad_(True)
            return self.leaf
        return out


def reproj_case(name, seed, bs=3, n_pix=10):
    """The keyframe re-projection blocks of a mapping iteration under bundle adjustment (camera tensors require grad,
    volsdf_train.py:521-528): flow (network.py:153-165) and patch warp with patch sizes 1 and 5 (:167-279) on 40x60 frames.
    Run twice on the reference: (A) as is -- forward tensors, total camera gradient; (B) with the rendered depth cut into a
    leaf (_CutDepth) -- d/d(rendered depth) and the DIRECT camera gradient of the masked-L1 terms (loss.py:106-111,136-142)."""
    coarse_grid, fine_grid, colour_grid = (4, 4, 8, 4, 8), (4, 32, 10, 8, 4), (4, 64, 10)
    samples = (10, 32, 6)
    H, W = _SmallDS.img_res
    g = torch.Generator().manual_seed(seed + 1)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 30.0
    K[0, 2], K[1, 2] = 29.5, 19.5
    K[0, 1] = 0.3                                  # skew
    idx = torch.randint(H * W, (bs, n_pix), generator=g)
    uv = torch.stack([(idx % W).float(), (idx // W).float()], -1)
    uv[0, 0] = torch.tensor([1.0, 1.0])            # patch partly outside the image
    uv[1, 1] = torch.tensor([float(W - 1), float(H - 2)])
    cam0 = torch.zeros(bs, 7)
    cam0[:, 0] = 1.0
    cam0[:, :4] += 0.03 * torch.randn(bs, 4, generator=g)
    cam0[:, 4:] = torch.tensor([0.1, 0.0, -0.2]) + 0.03 * torch.randn(bs, 3, generator=g)
    Kb = K[None].repeat(bs, 1, 1)
    full_rgb = torch.rand(bs, H * W, 3, generator=g)
    full_depth = 1.0 + 0.02 * torch.rand(bs, H * W, 1, generator=g)
    full_depth[:, : H * W // 3] += torch.rand(bs, H * W // 3, 1, generator=g)
    gt_rgb = torch.rand(bs * n_pix, 3, generator=g)
    edges = (torch.tensor([0, 1, 1, 2]), torch.tensor([1, 0, 2, 1]), torch.tensor([0, 10, 10, 20]), torch.tensor([10, 0, 20, 10]))
    gt_flow = (torch.rand(4, n_pix, 2, generator=g) - 0.5) * 6
    flow_mask = torch.rand(4, n_pix, generator=g) > 0.3
    rec = {"in_uv": uv, "in_cam": cam0, "in_K": Kb, "in_voxels": torch.zeros(64, 64, 64), "in_full_rgb": full_rgb,
           "in_full_depth": full_depth, "gt_rgb": gt_rgb, "in_idii": edges[0], "in_idjj": edges[1], "gt_flow": gt_flow,
           "gt_flow_mask": flow_mask, "meta_mode": "mapping", "meta_stage": "fine", "meta_color_stage": "highfreq",
           "meta_training": 1, "meta_samples": np.array(samples), "meta_coarse_grid": np.array(coarse_grid),
           "meta_fine_grid": np.array(fine_grid), "meta_colour_grid": np.array(colour_grid), "meta_img_res": np.array([H, W])}

    def reproj_terms(out):
        terms = []
        for ps, (gt_w, samp, mask, ray_mask) in out["warp_output"].items():
            terms.append((samp[mask] - gt_w[mask]).abs().mean())                              # loss.py:136-142
        fl = (out["flow"][flow_mask] - gt_flow[flow_mask]).abs().mean()                         # loss.py:106-111
        return terms, fl

    for run in ("A", "B"):
        model, conf = build_model(seed, coarse_grid, fine_grid, colour_grid, *samples, emb_scale=(0.05, 0.05, 0.5),
                                  warp=True, ds=_SmallDS())
        model.train(True)
        cam = cam0.clone().requires_grad_(True)
        pose = ref_general.get_camera_from_tensor(cam)
        pose.retain_grad()
        log = DrawLog()
        torch.manual_seed(seed + 3)
        cut = _CutDepth(bs * n_pix) if run == "B" else contextlib.nullcontext()
        with capture_draws(log), cut:
            out = model({"intrinsics": Kb, "uv": uv, "pose": pose}, torch.arange(bs),
                        {"full_rgb": full_rgb, "full_depth": full_depth, "edges": edges}, mode="mapping", stage="fine",
                        color_stage="highfreq", frame_idx=20)
        terms, fl = reproj_terms(out)
        if run == "A":
            rec["in_pose"] = pose.detach()
            rec["draw_t_rand"], rec["draw_extra_idx"] = log.draws[0][1], log.draws[1][1][: samples[2]]
            rec["draw_eik_idx"], rec["draw_eik_uniform"], rec["draw_eik_jitter"] = (log.draws[2][1], log.draws[3][1],
                                                                                     log.draws[4][1])
            for k, v in model.state_dict().items():
                rec["param_" + k] = v
            rec["out_flow"] = out["flow"]
            for ps, (gt_w, samp, mask, ray_mask) in out["warp_output"].items():
                rec[f"out_warp{ps}_gt"], rec[f"out_warp{ps}_sampled"], rec[f"out_warp{ps}_mask"] = gt_w, samp, mask
                if ray_mask is not None:
                    rec[f"out_warp{ps}_raymask"] = ray_mask
            rec["out_z_vals"], rec["out_rgb_values"], rec["out_depth_values"] = (out["z_vals"], out["rgb_values"],
                                                                                  out["depth_values"])
            rec["out_warp_terms"], rec["out_flow_term"] = torch.stack(terms), fl
            loss = (out["rgb_values"].reshape(-1, 3) - gt_rgb).abs().mean() + 0.5 * sum(terms) + 0.1 * fl
            loss.backward()
            rec["out_loss"], rec["grad_cam"], rec["grad_pose"] = loss, cam.grad, pose.grad
        else:
            leaf = cut.leaf
            assert leaf is not None
            rec["in_rendered_depth"] = leaf.detach().reshape(bs, n_pix)
            # one backward per term: d term / d depth and the direct d term / d pose
            for tag, term in [(f"warp{ps}", t) for ps, t in zip(out["warp_output"].keys(), terms)] + [("flow", fl)]:
                leaf.grad = None
                pose.grad = None
                term.backward(retain_graph=True)
                rec[f"grad_depth_{tag}"] = leaf.grad.reshape(bs, n_pix).clone()
                rec[f"grad_pose_{tag}"] = pose.grad.clone()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, float(rec["out_loss"]), [float(t) for t in rec["out_warp_terms"]], float(rec["out_flow_term"]),
          {k: tuple(v.shape) for k, v in rec.items() if k.startswith("grad_")})


def loss_case(name, seed, frame_idx, stage, bs=2, n=12, S=6, family="replica"):
    """SLAMLoss.forward (model/loss.py:113-233 + utils/MiDaS.py) with the shipped Replica weights
    (confs/replica/runconf_replica_1.conf:45-56; family "7scenes": confs/7scenes/runconf_7scenes_1.conf:46-58, smooth_weight 0.05;
    family "azure": confs/azure/runconf_azure_2.conf:46-59, assign_scale 15 -- it scales the first frame's monocular depth, loss.py:179-185)
    on random model outputs: every term and d loss / d output."""
    ref_loss = ref_shims.import_ref("model.loss")
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *shape: torch.rand(*shape, generator=g)
    leaf = lambda t: t.clone().requires_grad_(True)
    out = {"rgb_values": leaf(rnd(bs, n, 3)), "depth_values": leaf(rnd(bs, n, 1) * 2 + 0.2),
           "normal_map": leaf(rnd(bs, n, 3) - 0.5), "grad_theta": leaf((rnd(40, 3) - 0.5) * 3),
           "grad_theta_nei": leaf((rnd(40, 3) - 0.5) * 3), "flow": leaf((rnd(3, n, 2) - 0.5) * 8)}
    sdf = rnd(bs * n, S) - 0.4
    sdf[::5] = sdf[::5].abs()            # some rays never cross the surface -> masked out
    out["sdf"] = sdf
    warp = {}
    for ps in (1, 5):
        m = rnd(bs * n, ps * ps) > 0.3
        warp[ps] = (rnd(bs * n, ps * ps, 3), leaf(rnd(bs * n, ps * ps, 3)), m, m.any(-1))
    out["warp_output"] = warp
    gt = {"rgb": rnd(bs, n, 3), "depth": rnd(bs, n, 1) * 0.02, "normal": rnd(bs, n, 3) - 0.5,
          "gt_depth": rnd(bs, n, 1) * 3 * (rnd(bs, n, 1) > 0.2), "mask": (rnd(bs, n, 1) > 0.15).float(),
          "flow": (rnd(3, n, 2) - 0.5) * 8, "flow_mask": rnd(3, n) > 0.4}

    class DS:
        data_dir = {"replica": "../Datasets/processed/Replica", "7scenes": "../Datasets/processed/7Scenes",
                    "azure": "../Datasets/processed/Azure"}[family]
    smooth_weight = 0.05 if family == "7scenes" else 0.005
    extra = dict(assign_scale=15.0) if family == "azure" else {}        # confs/azure/runconf_azure_2.conf:47
    crit = ref_loss.SLAMLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, train_dataset=DS(), scan_id=1, **extra,
                             assign_scale_shift_init=True, smooth_weight=smooth_weight, warp_loss_type="l1", depth_weight=0.1,
                             normal_l1_weight=0.05, normal_cos_weight=0.05, flow_weight=0.001, warp_loss_weight=0.5)
    res = crit(out, gt, keyframe_list=None, frame_idx=frame_idx, stage=stage)
    res["loss"].backward()
    rec = {"meta_frame_idx": np.array(frame_idx), "meta_stage": np.array(stage), "meta_smooth_weight": np.array(smooth_weight),
           "meta_data_dir": np.array(DS.data_dir)}
    for k, v in out.items():
        if isinstance(v, torch.Tensor):
            rec["in_" + k] = v.detach()
            if v.grad is not None:
                rec["grad_" + k] = v.grad
    for ps, (a, b, m, rm) in warp.items():
        rec.update({f"in_warp{ps}_gt": a, f"in_warp{ps}_sampled": b.detach(), f"in_warp{ps}_mask": m,
                    f"in_warp{ps}_raymask": rm})
        if b.grad is not None:
            rec[f"grad_warp{ps}_sampled"] = b.grad
    for k, v in gt.items():
        rec["gt_" + k] = v
    for k, v in res.items():
        rec["out_" + k] = torch.as_tensor(float(v))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, {k: float(v) for k, v in res.items()})


def func_case(name, seed):
    """Function-level vectors for the rows that are plain PyTorch in the reference:
    rend_util.get_camera_params (skewed K), UniformSampler.near_far_from_cube, get_embedder(6)/(4), GridPredefineDensity,
    SLAMNetwork.volume_rendering, general.quad2rotation / get_camera_from_tensor -- inputs and the reference's outputs."""
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.rand(*s, generator=g)
    rec = {}
    # a17: camera 7-vector (non-unit quaternion) -> 4x4
    cam = torch.cat([torch.tensor([[1.0, 0, 0, 0]]).repeat(3, 1) + 0.2 * (rnd(3, 4) - 0.5), rnd(3, 3) - 0.5], -1)
    rec.update(a17_cam=cam, a17_pose=ref_general.get_camera_from_tensor(cam), a17_rot=ref_general.quad2rotation(cam[:, :4]))
    # a1: pixels -> rays, with skew
    uv, _, K = synth_inputs(seed, 3, 11)
    pose = rec["a17_pose"]
    dirs, loc = ref_rend.get_camera_params(uv, pose, K)
    rec.update(a1_uv=uv, a1_K=K, a1_pose=pose, a1_ray_dirs=dirs, a1_cam_loc=loc)
    # a2: cube intersection
    samplers = ref_shims.import_ref("model.ray_sampler")
    us = samplers.UniformSampler(1.0, 0.0, 16, 3.5, take_sphere_intersection=False) if False else None
    o = (rnd(40, 3) - 0.5) * 1.2
    d = torch.nn.functional.normalize(rnd(40, 3) - 0.5, dim=-1) * (0.5 + rnd(40, 1))
    d[0, 1] = 0.0                                    # an axis-parallel component: the +1e-15 guard
    o[1] = torch.tensor([1.5, 1.5, 1.5])             # outside, pointing away: no intersection
    d[1] = torch.tensor([1.0, 0.5, 0.2])

    class _U:
        near, far = 0.0, 3.5
    near, far = samplers.UniformSampler.near_far_from_cube(_U(), o.clone(), d.clone(), 1.0)
    rec.update(a2_o=o, a2_d=d, a2_near=near, a2_far=far)
    # a7: positional encodings
    emb = ref_shims.import_ref("model.embedder")
    x = (rnd(17, 3) - 0.5) * 2
    for m in (6, 4):
        fn, dim = emb.get_embedder(m, input_dims=3)
        rec[f"a7_pe{m}"] = fn(x)
    rec["a7_x"] = x
    # a11: beta from the visit counter, Laplace density
    dens = ref_shims.import_ref("model.density").GridPredefineDensity()
    dens.voxels = torch.poisson(torch.full((64, 64, 64), 30.0), generator=g)
    dens.voxel_res = 64
    pts = (rnd(50, 3) - 0.5) * 2.02                  # some |x_d| > 0.99
    sdf = (rnd(50, 1) - 0.5) * 0.2
    rec.update(a11_voxels=dens.voxels, a11_x=pts, a11_sdf=sdf, a11_beta=dens.get_beta(pts), a11_sigma=dens(sdf, x=pts))
    # a13: weights of one batch of rays
    model, conf = build_model(seed, (4, 4, 8, 4, 8), (4, 32, 10, 8, 4), (4, 64, 10), 10, 32, 6, (0.05, 0.05, 0.3))
    model.voxels = dens.voxels
    model.density.voxels = dens.voxels
    z = torch.sort(rnd(7, 9) * 2, dim=1).values
    sd = (rnd(7 * 9, 1) - 0.4) * 0.1
    xs = (rnd(7 * 9, 3) - 0.5) * 1.9
    rec.update(a13_z=z, a13_sdf=sd, a13_x=xs, a13_weights=model.volume_rendering(z, sd, xs))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, {k: tuple(v.shape) for k, v in rec.items()})


def feed_case(name, seed):
    """Per-iteration feed through the reference's own SLAMDataset.change_sampling_idx / __getitem__ /
    collate_fn (code/datasets/scene_dataset.py:214-287), on an instance whose image caches hold synthetic frames (its
    __init__ only reads files, so the object is allocated without it).  Three batches: mapping (2 frames, random pixels),
    tracking (1 frame; pixels drawn from the first tracking_total_pixels indices, :282-286), visualisation (whole image)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_scene_dataset", os.path.join(ref_shims.REF_CODE, "datasets",
                                                                                     "scene_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    H, W, Hedge, Wedge = 12, 20, 2, 3
    g = torch.Generator().manual_seed(seed)
    ds = object.__new__(mod.SLAMDataset)
    ds.img_res, ds.H, ds.W = (H, W), H, W
    ds.total_pixels = H * W
    ds.tracking_total_pixels = (H - 2 * Hedge) * (W - 2 * Wedge)
    ds.scene_scale = 2.5
    ds.sampling_idx = None
    i = torch.arange(H * W)
    ds.uv = torch.stack([(i % W).float(), (i // W).float()], -1)     # the pixel grid of scene_dataset.py:106-111
    ds.rgb_images, ds.mask_images, ds.depth_images, ds.normal_images, ds.gt_depth_images = {}, {}, {}, {}, {}
    ds.intrinsics_all, ds.est_pose_all = {}, {}
    rec = dict(meta_res=np.array([H, W, Hedge, Wedge]), meta_scene_scale=np.float64(ds.scene_scale))
    for idx in (3, 8):
        ds.rgb_images[idx] = torch.rand(H * W, 3, generator=g)
        ds.depth_images[idx] = torch.rand(H * W, 1, generator=g)
        ds.normal_images[idx] = torch.rand(H * W, 3, generator=g) * 2 - 1
        ds.gt_depth_images[idx] = torch.rand(H * W, 1, generator=g) * 4
        ds.mask_images[idx] = (torch.rand(H * W, 1, generator=g) > 0.2).float()
        K = torch.eye(4)
        K[0, 0] = K[1, 1] = 10.0 + idx
        K[0, 2], K[1, 2] = W / 2 - 0.5, H / 2 - 0.5
        ds.intrinsics_all[idx] = K
        ds.est_pose_all[idx] = torch.eye(4) + 0.01 * torch.randn(4, 4, generator=g)
        for k, t in (("rgb", ds.rgb_images), ("depth", ds.depth_images), ("normal", ds.normal_images),
                     ("gt_depth", ds.gt_depth_images), ("mask", ds.mask_images), ("intrinsics", ds.intrinsics_all),
                     ("pose", ds.est_pose_all)):
            rec[f"frame{idx}_{k}"] = t[idx]
    torch.manual_seed(seed)
    for tag, mode, frames, n in (("map", "mapping", [8, 3], 7), ("trk", "tracking", [3], 5), ("vis", "mapping", [8], -1)):
        ds.mode = mode
        ds.change_sampling_idx(n)
        indices, inp, gt = ds.collate_fn([ds[f] for f in frames])
        rec[f"{tag}_frames"] = np.array(frames)
        rec[f"{tag}_indices"] = indices
        if ds.sampling_idx is not None:
            rec[f"{tag}_sampling_idx"] = ds.sampling_idx
        for k, v in inp.items():
            rec[f"{tag}_in_{k}"] = v
        for k, v in gt.items():
            rec[f"{tag}_gt_{k}"] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **t2n(rec))
    print(name, sorted(k for k in rec if not k.startswith("frame")))


def rw_cases():
    full_case("full_tracking_rw", 16, "tracking", "fine", "highfreq", bs=1, n_pix=24, poisson=True, rand_v=0.04)
    full_case("full_mapping_rw", 17, "mapping", "fine", "highfreq", bs=2, n_pix=8, poisson=True, rand_v=0.04)
    full_case("full_mapping_rw_coarse", 18, "mapping", "coarse", "highfreq", bs=2, n_pix=8, rand_v=0.04)


def family_cases():
    """the second conf family (7-Scenes / Azure model subtree + 7-Scenes frame size and camera): tracking, both mapping schedules"""
    full_case("full_tracking_7scenes", 61, "tracking", "fine", "highfreq", bs=1, n_pix=24, poisson=True, family="7scenes")
    full_case("full_mapping_7scenes", 62, "mapping", "fine", "highfreq", bs=2, n_pix=8, poisson=True, family="7scenes")
    full_case("full_mapping_7scenes_coarse_base", 63, "mapping", "coarse", "base", bs=2, n_pix=8, family="7scenes")


def twin_cases():
    twin_case("twin_dense", 4, L=5, C=4, base=8, end=24, n_pts=128)
    twin_case("twin_dense_c8", 6, L=3, C=8, base=8, end=19, n_pts=96)
    twin_case("twin_dense_c2", 7, L=6, C=2, base=4, end=21, n_pts=96)


if __name__ == "__main__":
    if "--twin-only" in sys.argv:
        twin_cases()
        sys.exit(0)
    if "--rw-only" in sys.argv:
        rw_cases()
        sys.exit(0)
    if "--family-only" in sys.argv:
        family_cases()
        loss_case("loss_mapping_7scenes", 23, frame_idx=7, stage="fine", family="7scenes")
        loss_case("loss_mapping_azure_first_frame", 24, frame_idx=0, stage="fine", family="azure")
        sys.exit(0)
    if "--reproj-only" in sys.argv:
        reproj_case("reproj_blocks", 51)
        sys.exit(0)
    if "--feed-only" in sys.argv:
        feed_case("feed_batches", 41)
        sys.exit(0)
    func_case("func_rows", 31)
    if "--func-only" in sys.argv:
        sys.exit(0)
    loss_case("loss_mapping_first_frame", 21, frame_idx=0, stage="coarse")
    loss_case("loss_mapping_fine", 22, frame_idx=7, stage="fine")
    if "--loss-only" in sys.argv:
        sys.exit(0)
    warp_case("full_mapping_warp", 15)
    reproj_case("reproj_blocks", 51)
    encoder_case("enc_coarse", 1, L=4, C=8, base=8, end=8, logmap=19, n_pts=64)
    encoder_case("enc_fine", 2, L=8, C=4, base=4, end=40, logmap=10, n_pts=64)
    encoder_case("enc_colour", 3, L=16, C=2, base=4, end=128, logmap=11, n_pts=64)
    twin_cases()
    feed_case("feed_batches", 41)
    sparse_colour_case("enc_colour_real_sparse", 5, n_pts=24)
    full_case("full_tracking", 10, "tracking", "fine", "highfreq", bs=1, n_pix=24)
    full_case("full_tracking_poisson", 11, "tracking", "fine", "highfreq", bs=1, n_pix=16, poisson=True)
    full_case("full_mapping", 12, "mapping", "fine", "highfreq", bs=2, n_pix=8, poisson=True)
    full_case("full_mapping_coarse_base", 13, "mapping", "coarse", "base", bs=2, n_pix=8)
    full_case("full_vis_eval", 14, "vis", "fine", "highfreq", bs=1, n_pix=16, training=False)
    rw_cases()
    family_cases()
    loss_case("loss_mapping_7scenes", 23, frame_idx=7, stage="fine", family="7scenes")
    loss_case("loss_mapping_azure_first_frame", 24, frame_idx=0, stage="fine", family="azure")
