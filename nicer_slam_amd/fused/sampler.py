"""Fused hierarchical sampler: thin wrapper over nsa_sampler_sdf + nsa_sample_rays (include/nicer_slam_amd.h,
Section 2).  Replaces ImportantSampler.get_z_vals of the composed engine (reference code/model/ray_sampler.py:90-166)
with two kernel launches; no gradient flows through the sampler (the reference runs it on detached rays under
no_grad, ray_sampler.py:38-39,101-102)."""
import ctypes
import os

import numpy as np
import torch

from .._native import lib, check, GridDesc
from ..hashencoder.backend import _offsets_host, _timed
from . import pack

_LIN = {}


def _linspace(n, device):
    key = (n, str(device))
    if key not in _LIN:
        _LIN[key] = torch.linspace(0.0, 1.0, steps=n, device=device)
    return _LIN[key]


PRECISIONS = ("fp32", "bf16", "bf16_colour")


def precision_of(model, which):
    """nsa_grid_t.precision for the SDF networks (which = "sdf") or the colour network ("colour") under
    ``model.mlp_precision``: "fp32" (default: fp32-faithful GEMMs everywhere), "bf16" (every MLP takes bf16 operands)
    or "bf16_colour" (bf16 colour MLP, fp32 SDF head -- BASELINE configs[4])."""
    mode = getattr(model, "mlp_precision", "fp32")
    if mode not in PRECISIONS:
        raise ValueError(f"mlp_precision must be one of {PRECISIONS}, got {mode!r}")
    return 1 if mode == "bf16" or (mode == "bf16_colour" and which == "colour") else 0


# Tiling of the SDF-network kernels per use (nsa_grid_t.tile): 16 = quad tiling (16 points per wave, four lanes per point,
# csrc/render_sdfnet4.hip / render_sampler4.hip), 32 = 32-point tiling (lane pair per point, csrc/render_sdfnet.hip /
# render_sampler.hip).  Both compute the same numbers (tests/test_tiling_gpu.py); the defaults are what measured fastest on
# MI355X (profiles/r02_*): the fine network (three hidden layers: 256 .. 500 registers per lane at 32 points) runs the quad
# tiling, the coarse network and the sampler's SDF-only pass the 32-point one -- except the coarse network's MAP backward
# (parameter gradients: one wave per SIMD at 32 points, two in quad form: 969 -> 748 us per launch at 8192 rays).
# NSA_SDF_TILE=16|32 or ``model.sdf_tile`` force one tiling everywhere.
# "sampler": 64 = the 32-point tiling with TWO point tiles per wave: one weight-fragment stream feeds both tiles and the
# operand split of one overlaps the matrix instructions of the other; bit-identical results, 2 waves per SIMD instead of 3,
# 208-214 -> 206 us (profiles/r02_ab_experiments.txt r3b).
# "coarse_pair": the coarse network inside nsa_sdfnet_forward_pair (both networks' forward in one launch; quad tiling only).
# "sampler_large" (>= 4096 rays, the mapping batch): the persistent quad sampler (16) was the faster form there in round 2 (1739 -> 1645 us
# at 8192 rays); since round 3's work on the two-tile 32-point kernel that one is: 1517 vs 1677 us (profiles/r04_ab_experiments.txt r4u).
# "sampler_small" (<= 256 rays: the per-GPU share of an 8- / 4-GPU strong-scaling run of the 1024-ray batch): ONE tile per wave -- with at
# most 2.5 waves per SIMD the launch is one round of waves and a wave's life is what counts: 0.1872 -> 0.1812 ms per iteration at 128 rays,
# 0.2103 -> 0.2060 at 256; from 512 rays up the two-tile form wins again (0.3193 vs 0.3346; profiles/r06_ab_experiments.txt r6c).
DEFAULT_TILES = {"coarse": 32, "fine": 16, "sampler": 64, "coarse_map": 16, "sampler_large": 64, "coarse_pair": 16, "sampler_small": 32}
FWD_PAIR = True          # False: two forward launches (module attribute: the bit-identity tests and A/B runs flip it)
# 1: the colour backward and the coarse SDF backward of a data-path backward as ONE launch (nsa_colour_coarse_backward; 32-point tiling of
# the coarse network).  False: two launches.
COLOUR_COARSE_BWD = True
# 1: the tracker's colour forward also runs the ray's composite + L1 + composite backward (nsa_colour_forward_track) when a ray is exactly
# one workgroup of the colour forward (128 samples per ray).  False: nsa_colour_forward + nsa_composite_track.
COLOUR_FWD_TRACK = True
SAMPLER_LARGE_RAYS = 4096
SAMPLER_SMALL_RAYS = 256


def sampler_use(n_rays):
    """the DEFAULT_TILES key of the SDF-only sampler pass for a batch of n_rays"""
    return "sampler_large" if n_rays >= SAMPLER_LARGE_RAYS else "sampler_small" if n_rays <= SAMPLER_SMALL_RAYS else "sampler"
_FORCE = int(os.environ.get("NSA_SDF_TILE", "0"))
_FORCE_SAMPLER = 0       # A/B runs of the sampler pass alone set this to 16 | 32 | 64 (96 / 97: the wave-specialised experiment builds)


def tile_of(model, which):
    """``which``: "coarse" / "fine" (composite-pass kernels of that network), "coarse_map" (the coarse network's MAP backward),
    "coarse_pair" (the coarse network inside the paired forward), "sampler" / "sampler_large" (SDF-only pass, both networks; by
    batch size)."""
    t = int(getattr(model, "sdf_tile", 0) or _FORCE or DEFAULT_TILES[which])
    if which.startswith("sampler") and _FORCE_SAMPLER and not getattr(model, "sdf_tile", 0):
        t = _FORCE_SAMPLER
    if t in (64, 96, 97) and not which.startswith("sampler"):
        t = 32                                 # (both are forms of the 32-point tiling's sampler pass)
    if t not in (16, 32) and not (t in (64, 96, 97) and which.startswith("sampler")):
        raise ValueError(f"sdf_tile must be 16 or 32 (sampler only: 64 = 32-point tiling with two tiles per wave, 96 / 97 = "
                         f"wave-specialised 32-point forms: matrix waves on loan / systolic layer engines), got {t}")
    return t


def forward_pair_ok(model):
    """The composite pass runs both SDF networks' forward as ONE launch (nsa_sdfnet_forward_pair) when the quad tiling is in use."""
    return FWD_PAIR and tile_of(model, "fine") == 16 and tile_of(model, "coarse_pair") == 16


def grid_desc(net_or_enc, divide_factor, n_hidden, precision=0, tile=0):
    enc = net_or_enc
    off = _offsets_host(enc.offsets)
    d = GridDesc(enc.embeddings.data_ptr(), off.data_ptr(), enc.num_levels, enc.level_dim,
                 float(np.log2(enc.per_level_scale)), enc.base_resolution, float(divide_factor), n_hidden, precision, tile)
    return d, (off, enc.embeddings)


def sdf_grid_desc(model, which, use=None):
    """nsa_grid_t of the coarse / fine SDF network under the model's precision and tiling settings; ``use`` = "sampler"
    selects the tiling of the SDF-only pass (default: the composite-pass tiling of that network)."""
    net = getattr(model.implicit_network, which)
    return grid_desc(net.encoding, net.divide_factor, net.num_layers - 2, precision_of(model, "sdf"), tile_of(model, use or which))


def supported(model):
    """Configurations the compiled kernels cover: the architecture shared by all 23 shipped run configs."""
    imp = model.implicit_network
    c, f = imp.coarse, imp.fine

    def ok(net, L, C, nh):
        e = net.encoding
        return (e.num_levels == L and e.level_dim == C and e.input_dim == 3 and net.num_layers - 2 == nh
                and net.multires == 6 and not net.skip_in and not net.clamp and net.use_grid_feature
                and all(d == 64 for d in net.dims[1:-1]) and net.dims[-1] == 65
                and net.encoding.embeddings.dtype == torch.float32)
    return (ok(c, 4, 8, 1) and ok(f, 8, 4, 3) and model.density_method == "volsdf_gridpredefined"
            and model.feature_vector_size == 64 and model.voxels.is_cuda and sample_counts_ok(model.ray_sampler))


MAX_S = 256                      # = MAX_S of csrc/render_sampler.hip / MAX_PER*64 of csrc/render_composite.hip
LDS_BYTES = 160 * 1024           # bound on E kept from the wave-per-ray k_sample_rays of rounds 1-2: 4 x (3 E + MAX_S) floats


def sample_counts_ok(samp):
    """Sample-count limits compiled into the per-ray kernels (nsa_sample_rays / nsa_composite_* return NSA_EBADARG beyond
    them): S = N_samples + 2 + N_samples_extra <= 256, E = N_samples_eval >= 2, four rays' pdf/cdf/z/sort buffers in 160 KiB
    of LDS, and at most E extras.  Outside them ``engine='auto'`` uses the composed engine."""
    S = samp.N_samples + 2 + samp.N_samples_extra
    E = samp.N_samples_eval
    return (1 <= S <= MAX_S and E >= 2 and (3 * E + MAX_S) * 4 * 4 <= LDS_BYTES and 0 <= samp.N_samples_extra <= E
            and samp.N_samples >= 1)


def packed_sdf(model, which, detach=True, use=None):
    """Packed parameters of the coarse/fine SDF MLP in the layout of the tiling used (``use``: see sdf_grid_desc), cached on
    the parameters' version counters."""
    net = getattr(model.implicit_network, which)
    params = net.mlp_parameters()
    tile = tile_of(model, use or which)
    key = tuple((p.data_ptr(), p._version) for p in params)
    cache = model.__dict__.setdefault("_fused_pack", {})
    hit = cache.get((which, tile))
    if detach and hit is not None and hit[0] == key:
        return hit[1]
    with torch.set_grad_enabled(not detach):
        packed = pack.pack_sdf_net4(net) if tile == 16 else pack.pack_sdf_net(net)
    if detach:
        cache[(which, tile)] = (key, packed)
    return packed


def sampler_sdf(model, rays_o, rays_d, t_rand):
    """-> z[R,E], sdf[R,E], far[R] (coarse stage)."""
    samp = model.ray_sampler
    us = samp.uniform_sampler
    R, E = rays_o.shape[0], samp.N_samples_eval
    dev = rays_o.device
    imp = model.implicit_network
    use = sampler_use(R)
    gc, keep_c = sdf_grid_desc(model, "coarse", use)
    gf, keep_f = sdf_grid_desc(model, "fine", use)
    pc, pf = packed_sdf(model, "coarse", use=use), packed_sdf(model, "fine", use=use)
    z = torch.empty(R, E, device=dev)
    sdf = torch.empty(R, E, device=dev)
    far = torch.empty(R, device=dev)
    t_lin = _linspace(E, dev)
    rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
    if t_rand is not None:
        t_rand = t_rand.contiguous()
        assert t_rand.shape == (R, E) and t_rand.dtype == torch.float32
    # algorithmic bytes: both SDF grids, 2^3 corners x C x 4 B per level per point
    with _timed("k_sampler_sdf", R * E * (4 * 8 * 8 * 4 + 8 * 8 * 4 * 4)):
        check(lib.nsa_sampler_sdf(rays_o.data_ptr(), rays_d.data_ptr(), R, E, t_lin.data_ptr(),
                                  t_rand.data_ptr() if t_rand is not None else None, float(us.near),
                                  float(us.scene_bounding_sphere), float(us.far), ctypes.byref(gc), ctypes.byref(gf),
                                  pc.data_ptr(), pf.data_ptr(), z.data_ptr(), sdf.data_ptr(), far.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream))
    return z, sdf, far


def sample_rays(model, rays_o, rays_d, z, sdf, far, extra_idx, eik_idx):
    """-> z_vals[R,S] sorted, z_eik[R,1] (importance stage)."""
    samp = model.ray_sampler
    R, E = z.shape
    N = samp.N_samples
    n_extra = 0 if extra_idx is None else int(extra_idx.numel())
    S = N + 2 + n_extra
    dev = z.device
    z_vals = torch.empty(R, S, device=dev)
    z_eik = torch.empty(R, device=dev)
    u_lin = _linspace(N, dev)
    ex = extra_idx.to(torch.int32).contiguous() if n_extra else None
    ek = eik_idx.to(torch.int32).contiguous() if eik_idx is not None else None
    vox = model.voxels.contiguous()
    with _timed("k_sample_rays", R * E * 8):
        check(lib.nsa_sample_rays(rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), sdf.data_ptr(), far.data_ptr(),
                                  vox.data_ptr(), model.voxel_res, R, E, N, u_lin.data_ptr(),
                                  ex.data_ptr() if n_extra else None, n_extra, float(samp.near),
                                  ek.data_ptr() if ek is not None else None, z_vals.data_ptr(),
                                  z_eik.data_ptr() if ek is not None else None, torch.cuda.current_stream().cuda_stream))
    return z_vals, (z_eik.unsqueeze(-1) if ek is not None else None)


# The sampler's draws come from the engine's own Philox4x32-10 stream (nsa_draw: jitter, permutation picks and eikonal indices in
# one launch, state advanced on the device) -- seeded from torch's generator when the model first draws, so torch.manual_seed
# still fixes the run.  OWN_RNG = False: torch.rand + nsa_draw_picks (inside a captured graph torch's graph-safe generator adds four
# small launches in front of every replay).
OWN_RNG = True


def draw_state(model, stream_key=0):
    """{seed, call number, ticket, -} of nsa_draw on the model's device.  Created on first use: call it (or run one iteration)
    BEFORE capturing a graph around the sampler.

    ``stream_key``: one state PER CONCURRENT CALLER.  The kernel advances the state itself (the last workgroup of a launch resets
    the ticket and bumps the call number), so two nsa_draw launches that may run at the same time -- the ray chunks of
    KernelTracker(chunks > 1) on their forked streams -- must not share one: the ticket would count both launches' arrivals, bump
    the call number early and leave picks ranked against keys of two different calls.  Chunk c (first row ``lo``) therefore draws
    from its own state, seeded seed ^ mix(lo): independent jitter per chunk, still a function of torch.manual_seed."""
    states = model.__dict__.setdefault("_draw_states", {})
    dev = model.voxels.device
    st = states.get(stream_key)
    if st is None or st.device != dev:
        base = model.__dict__.get("_draw_seed")
        if base is None:
            base = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))    # torch's CPU generator: follows torch.manual_seed
            model.__dict__["_draw_seed"] = base
        seed = (base ^ ((int(stream_key) * 0x9E3779B97F4A7C15) & (2 ** 62 - 1))) if stream_key else base
        st = torch.tensor([seed, 0, 0, 0], dtype=torch.int64).to(dev)
        states[stream_key] = st
        if stream_key == 0:
            model.__dict__["_draw_state"] = st
    return st


def own_draws(model):
    """True when get_z_vals would make its draws with nsa_draw (training, no pinned draws, E <= 1024, the engine's own generator)."""
    return bool(model.training and model.draws is None and model.ray_sampler.N_samples_eval <= 1024 and OWN_RNG)


def get_z_vals(model, ray_dirs, cam_loc, need_eik=True, rows=None, drawn=None):
    """Drop-in for ImportantSampler.get_z_vals (fused engine).  ``need_eik=False`` (tracking) skips the near-surface
    eikonal sample, which only mapping consumes.  ``rows = (lo, hi)``: these rays are rows lo..hi of a larger batch (a chunk of
    KernelTracker) -- pinned per-ray draws (``model.draws``, tests) are sliced accordingly.  ``drawn = (t_rand [R,E], extra_idx
    [n_extra] int32)``: this call's draws have been made already (nsa_track_begin_draw: the tracker's head launch) -- only when
    own_draws(model) and not need_eik."""
    samp = model.ray_sampler
    rays_d = ray_dirs.detach().contiguous()
    rays_o = cam_loc.detach().contiguous()
    R, E = rays_d.shape[0], samp.N_samples_eval
    n_extra = samp.N_samples_extra
    S = samp.N_samples + 2 + n_extra
    dev = rays_d.device
    if drawn is not None:
        if need_eik or not own_draws(model):
            raise RuntimeError("get_z_vals(drawn=...): only for the engine's own draws without eikonal picks")
        t_rand, extra = drawn
        if t_rand.shape != (R, E) or (n_extra > 0 and (extra is None or extra.numel() != n_extra)):
            raise ValueError("get_z_vals(drawn=...): draw buffers of the wrong shape")
        z, sdf, far = sampler_sdf(model, rays_o, rays_d, t_rand)
        return sample_rays(model, rays_o, rays_d, z, sdf, far, extra if n_extra > 0 else None, None)
    if model.training and model.draws is None and E <= 1024 and OWN_RNG:
        # fast path: every draw of this call from ONE launch of the engine's own counter-based generator (nsa_draw)
        t_rand = torch.empty(R, E, device=dev)
        extra = torch.empty(n_extra, device=dev, dtype=torch.int32) if n_extra > 0 else None
        eik_idx = torch.empty(R, device=dev, dtype=torch.int32) if need_eik else None
        with _timed("k_draw", R * E * 4):
            check(lib.nsa_draw(draw_state(model, rows[0] if rows is not None else 0).data_ptr(), R * E, t_rand.data_ptr(), E, n_extra,
                               R, S, extra.data_ptr() if extra is not None else None,
                               eik_idx.data_ptr() if eik_idx is not None else None, torch.cuda.current_stream().cuda_stream))
        z, sdf, far = sampler_sdf(model, rays_o, rays_d, t_rand)
        return sample_rays(model, rays_o, rays_d, z, sdf, far, extra, eik_idx)
    if model.training and model.draws is None and E <= 1024:
        # the same with torch's generator: ONE device rand for every draw of this call + one kernel for the integer picks
        u = torch.rand(R * E + E + R, device=dev)
        z, sdf, far = sampler_sdf(model, rays_o, rays_d, u[:R * E].view(R, E))
        extra = torch.empty(n_extra, device=dev, dtype=torch.int32) if n_extra > 0 else None
        eik_idx = torch.empty(R, device=dev, dtype=torch.int32) if need_eik else None
        if extra is not None or eik_idx is not None:
            check(lib.nsa_draw_picks(u[R * E:].data_ptr(), E, n_extra, R, S, extra.data_ptr() if extra is not None else None,
                                     eik_idx.data_ptr() if eik_idx is not None else None,
                                     torch.cuda.current_stream().cuda_stream))
        return sample_rays(model, rays_o, rays_d, z, sdf, far, extra, eik_idx)
    t_rand = model.draw("t_rand", (R, E)) if model.training else None
    if t_rand is not None and rows is not None and t_rand.shape[0] != R:
        t_rand = t_rand[rows[0]:rows[1]]
    z, sdf, far = sampler_sdf(model, rays_o, rays_d, t_rand)
    if n_extra > 0:
        if model.training:
            extra = model.draw("extra_idx", (E, n_extra))
        else:
            extra = torch.linspace(0, E - 1, n_extra, device=z.device).long()
    else:
        extra = None
    eik_idx = model.draw("eik_idx", (S, R)) if need_eik else None
    if eik_idx is not None and rows is not None and eik_idx.shape[0] != R:
        eik_idx = eik_idx[rows[0]:rows[1]]
    return sample_rays(model, rays_o, rays_d, z, sdf, far, extra, eik_idx)
