"""The tracking forward / backward of an UNMODIFIED caller as two cached hipGraphs behind one autograd.Function.

The reference's tracking loop (code/training/volsdf_train.py:406-443) is eager: camera tensor -> get_camera_from_tensor ->
model(input, ..., mode="tracking") -> loss -> loss.backward() -> optimizer.step().  Driven that way the fused engine was host-bound
(bench.py `dropin.pose_only_eager`, round 3: 1.045 ms per iteration against 0.62 ms of device time): eleven launches of ours with
their ctypes glue, ~25 buffer allocations, pack-cache lookups and the autograd bookkeeping of four Functions per iteration.

Here SLAMNetwork.forward(mode="tracking") -- same signature, same dict, same gradients on ``input["pose"]`` -- runs

    copy(pose, uv) -> REPLAY [rays -> draws -> sampler SDF pass -> per-ray sampling -> SDF pair forward -> colour forward -> composite]
    ... caller's loss ...
    copy(d loss / d rgb_values) -> REPLAY [composite backward -> colour backward -> SDF backward x2 -> ray sums -> pose backward]

with both graphs captured on the second call of a given shape (the first runs eagerly and warms allocator, packs and lazy state)
and keyed on everything a replay reads through a captured ADDRESS: ray count, stage / colour stage, tilings, precision, the three
tables, the visit counter.  The packed MLP snapshots are owned by the cache and re-packed IN PLACE when a parameter version changes
(a mapping step between two tracked frames), exactly as KernelTracker does.  Cotangent patterns other than "rgb_values only" (the
tracking objective, loss.py:131) run the backward kernels eagerly.

Aliasing contract (round 5).  The per-ray results a caller's loop keeps across iterations -- rgb_values, depth_values, normal_map,
entropy: what the reference's loop logs, compares and renders from (volsdf_train.py:417-446) -- are returned as FRESH tensors, as
the eager path returns them (one nsa_copy_segments launch out of the static buffers into one new allocation per call).  The five
large per-sample tensors (weights, sdf, rgb, z_vals, depth_vals: [R,S]-sized, 3.7 MB per call) stay views of the graph's static
buffers, valid until the next forward(mode="tracking") on the same model; NSA_TRACK_CLONE=all clones those too, NSA_TRACK_CLONE=none
restores round 4's all-views behaviour.  A backward through the outputs of an earlier forward raises instead of returning another
iteration's gradient.

NSA_TRACK_GRAPH=0 keeps the eager Functions (A/B runs)."""
import os

import torch

from .._native import lib, check
from ..hashencoder.backend import _timed

ENABLED = os.environ.get("NSA_TRACK_GRAPH", "1") != "0"
CLONE = os.environ.get("NSA_TRACK_CLONE", "small")      # small | all | none (see the aliasing contract above)


# ---- packed MLP snapshots owned by a long-lived consumer (a captured graph reads their addresses) ---------------------------------
def pack_specs(model, n_rays, stage):
    """(cache key, network, use) of every packed block a tracking iteration over ``n_rays`` rays reads (fused/sampler.py::tile_of)."""
    from . import sampler as fs
    use = "sampler_large" if n_rays >= fs.SAMPLER_LARGE_RAYS else "sampler"
    specs = []
    for which in ("coarse", "fine"):
        for u in (use, None):
            specs.append(((which, fs.tile_of(model, u or which)), which, u))
    if stage != "coarse" and fs.forward_pair_ok(model):       # the paired forward reads the coarse net's quad pack
        specs.append((("coarse", fs.tile_of(model, "coarse_pair")), "coarse", "coarse_pair"))
    return specs


def ensure_packs(model, owned, specs):
    """Fresh packed blocks for every tiling in ``specs`` (+ the colour net) on the current stream, RE-PACKED IN PLACE into the
    tensors ``owned`` already holds: a captured graph keeps reading the same addresses after a parameter update.  (A plain cache
    miss would allocate a new snapshot and free the captured one while the graph kept replaying on stale -- possibly recycled --
    memory.)"""
    from . import render as fr, sampler as fs
    cache = model.__dict__.setdefault("_fused_pack", {})
    jobs = [(k, (lambda w=w, u=u: fs.packed_sdf(model, w, use=u)), getattr(model.implicit_network, w).mlp_parameters())
            for k, w, u in specs]
    jobs.append(("colour", lambda: fr.packed_colour(model), model.rendering_network.mlp_parameters()))
    for k, pack_fn, params in jobs:
        key = tuple((p.data_ptr(), p._version) for p in params)
        mine = owned.get(k)
        if mine is not None and mine[0] == key and cache.get(k, (None, None))[1] is mine[1]:
            continue
        if mine is not None:
            cache.pop(k, None)                     # force a re-pack, then move it into the tensor the graph knows
            fresh = pack_fn()
            mine[1].copy_(fresh)
            cache[k] = (key, mine[1])
            owned[k] = (key, mine[1])
        else:
            owned[k] = (key, pack_fn())
            cache[k] = owned[k]


# ---- the cached pair of graphs ---------------------------------------------------------------------------------------------------
def _key(model, R, stage, color_stage):
    """Everything a replay reads through a captured ADDRESS or a compiled choice.  (The intrinsics are NOT in it: the reference's
    loop hands over a fresh tensor from its data loader every iteration -- they are copied into a static buffer like the pixels.)"""
    from . import sampler as fs
    imp, rn, rs = model.implicit_network, model.rendering_network, model.ray_sampler
    # ... and every launch SCALAR a capture bakes in: the sampler's range and the scene / counter geometry
    return (R, stage, color_stage, getattr(model, "mlp_precision", "fp32"), getattr(model, "sdf_tile", 0), _tiles_key(fs),
            model.voxels.data_ptr(), imp.coarse.encoding.embeddings.data_ptr(), imp.fine.encoding.embeddings.data_ptr(),
            rn.encoding.embeddings.data_ptr(), rs.N_samples, rs.N_samples_eval, rs.N_samples_extra,
            float(getattr(rs, "near", 0.0)), float(getattr(rs, "far", 0.0)), float(getattr(rs, "scene_bounding_sphere", 0.0)),
            float(model.scene_bounding_sphere), int(model.voxel_res), bool(model.white_bkgd))


def _tiles_key(fs):
    return tuple(sorted(fs.DEFAULT_TILES.items()))


def usable(model, mode, fused_kind, input, ground_truth):
    """The calls this cache serves: training-mode tracking on the fused data path with the engine's own draws, one camera."""
    if not (ENABLED and mode == "tracking" and fused_kind == "data" and model.training and model.draws is None):
        return False
    from . import sampler as fs
    pose, uv, K = input["pose"], input["uv"], input["intrinsics"]
    return (fs.OWN_RNG and torch.is_grad_enabled() and pose.requires_grad and pose.is_cuda and pose.dim() == 3
            and pose.shape[0] == 1 and pose.shape[1] == 4 and uv.dim() == 3 and uv.shape[0] == 1 and uv.dtype == torch.float32
            and K.is_cuda and K.shape[-2:] == (4, 4) and "edges" not in ground_truth and not model.white_bkgd
            and model.ray_sampler.N_samples_eval <= 1024 and not torch.cuda.is_current_stream_capturing())


class TrackingGraph:
    def __init__(self, model, R, stage, color_stage):
        dev = model.voxels.device
        self.model, self.R, self.stage, self.color_stage = model, R, stage, color_stage
        self.pose_s = torch.zeros(1, 4, 4, device=dev)
        self.uv_s = torch.zeros(1, R, 2, device=dev)
        self.K = torch.zeros(1, 4, 4, device=dev)
        # every MLP parameter a packed block is built from: their version counters say when to re-pack (cheap per-call check)
        self._mlp_params = (list(model.implicit_network.coarse.mlp_parameters()) + list(model.implicit_network.fine.mlp_parameters())
                            + list(model.rendering_network.mlp_parameters()))
        self._mlp_stamp = None
        self.g_rgbv_s = torch.zeros(R, 3, device=dev)
        from .._native import CopySeg
        self._segs = (CopySeg * 3)(CopySeg(self.pose_s.data_ptr(), None, 16), CopySeg(self.uv_s.data_ptr(), None, 2 * R),
                                   CopySeg(self.K.data_ptr(), None, 16))
        self.packs = {}
        self.specs = pack_specs(model, R, stage)
        self.calls = 0
        self.serial = 0
        self.fwd_graph = self.bwd_graph = None
        self.out = self.g_pose = None
        self._out_segs = None

    # the two launch sequences (no autograd inside: plain C-ABI launches on the current stream)
    def _forward_body(self):
        from . import render as fr, sampler as fs
        model, R, dev = self.model, self.R, self.pose_s.device
        st = torch.cuda.current_stream().cuda_stream
        rays_o, rays_d, ds = (torch.empty(R, 3, device=dev), torch.empty(R, 3, device=dev), torch.empty(R, device=dev))
        samp = model.ray_sampler
        E, n_extra = samp.N_samples_eval, samp.N_samples_extra
        if fs.own_draws(model) and os.environ.get("NSA_TRACK_DRAW_IN_BEGIN", "1") != "0":
            # the sampler's draws ride in the ray-lifting launch (one graph node less); tracking consumes no eikonal sample
            t_rand = torch.empty(R, E, device=dev)
            extra = torch.empty(n_extra, device=dev, dtype=torch.int32) if n_extra > 0 else None
            with _timed("k_rays_fwd", R * 32):
                check(lib.nsa_rays_forward_draw(self.uv_s.data_ptr(), self.pose_s.data_ptr(), self.K.data_ptr(), 1, R, rays_o.data_ptr(),
                                                rays_d.data_ptr(), ds.data_ptr(), fs.draw_state(model, 0).data_ptr(), R * E,
                                                t_rand.data_ptr(), E, n_extra, samp.N_samples + 2 + n_extra,
                                                None if extra is None else extra.data_ptr(), st))
            z_vals, z_eik = fs.get_z_vals(model, rays_d, rays_o, need_eik=False, drawn=(t_rand, extra))
        else:
            with _timed("k_rays_fwd", R * 32):
                check(lib.nsa_rays_forward(self.uv_s.data_ptr(), self.pose_s.data_ptr(), self.K.data_ptr(), 1, R, rays_o.data_ptr(),
                                           rays_d.data_ptr(), ds.data_ptr(), st))
            z_vals, z_eik = fs.get_z_vals(model, rays_d, rays_o)
        b = fr.composite_forward_raw(model, rays_o, rays_d, z_vals, self.stage, True)
        # the forward's output dict (network.py:147-151, 281-300, 338-345) inside the same graph: five small torch launches that
        # would otherwise be dispatched, and recorded by autograd, on every call of the caller's loop
        rot = self.pose_s[:, :3, :3]
        final = dict(rgb_values=b["rgb_values"].view(1, R, 3), depth_values=(ds * b["depth"]).view(1, R, 1),
                     normal_map=torch.matmul(b["nmap"].view(1, R, 3), rot), entropy=b["entropy"].mean(),
                     depth_vals=z_vals * ds.view(R, 1))
        self.out = dict(b=b, rays_o=rays_o, rays_d=rays_d, ds=ds, z_vals=z_vals, z_eik=z_eik, final=final)

    def _backward_body(self, g_rgbv, g_depth=None, g_nmap=None, g_ent=None, g_w=None):
        from . import render as fr
        o = self.out
        g_o, g_d = fr.composite_backward_raw(self.model, o["rays_o"], o["rays_d"], o["z_vals"], o["b"], self.stage, self.color_stage,
                                             g_rgbv, g_depth, g_nmap, g_ent, g_w)
        g_pose = torch.empty(1, 4, 4, device=g_o.device)
        with _timed("k_rays_pose_bwd", self.R * 24):
            check(lib.nsa_rays_pose_backward(self.uv_s.data_ptr(), self.pose_s.data_ptr(), self.K.data_ptr(), 1, self.R,
                                             g_o.data_ptr(), g_d.data_ptr(), g_pose.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return g_pose

    def _copy_inputs(self, pose, uv, K):
        """pose, uv, K of this call into the graph's static buffers: ONE launch (nsa_copy_segments) when they are plain float32 device
        tensors, torch copies otherwise (strided / other dtype)."""
        R = self.R
        ok = lambda t, n: t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= n
        if ok(pose, 16) and ok(uv, 2 * R) and ok(K, 16) and uv.numel() == 2 * R:
            segs = self._segs
            segs[0].src, segs[1].src, segs[2].src = pose.data_ptr(), uv.data_ptr(), K.data_ptr()
            check(lib.nsa_copy_segments(segs, 3, torch.cuda.current_stream().cuda_stream))
        else:
            self.pose_s.copy_(pose.detach().reshape(1, 4, 4))
            self.uv_s.copy_(uv.detach())
            self.K.copy_(K.detach().reshape(-1, 4, 4)[:1])

    def forward(self, pose, uv, K):
        with torch.no_grad():
            self._copy_inputs(pose, uv, K)
            stamp = [(p.data_ptr(), p._version) for p in self._mlp_params]
            cache = self.model.__dict__.get("_fused_pack", {})
            # first call, a mapping step moved the MLPs, or another consumer replaced a cache entry this graph reads through a
            # captured address (the entry must still BE the tensor this graph owns): re-pack in place
            if stamp != self._mlp_stamp or any(cache.get(k, (None, None))[1] is not v[1] for k, v in self.packs.items()):
                ensure_packs(self.model, self.packs, self.specs)
                self._mlp_stamp = stamp
            self.calls += 1
            if self.fwd_graph is not None:
                self.fwd_graph.replay()
            elif self.calls == 1 or os.environ.get("NSA_TRACK_GRAPH") == "eager":
                self._forward_body()                               # warm-up: allocator pools, lazy state, the draw state
            else:
                self._backward_body(self.g_rgbv_s)                  # (warm the backward's pools on the eager forward's buffers)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward_body()
                self.fwd_graph = g
                gb = torch.cuda.CUDAGraph()                         # the backward of the tracking objective reads the forward's
                with torch.cuda.graph(gb):                          # static buffers: capture it now, on this thread
                    self.g_pose = self._backward_body(self.g_rgbv_s)
                self.bwd_graph = gb
                g.replay()
        self.serial += 1
        return self.out

    def fresh_small_outputs(self):
        """rgb_values [1,R,3], depth_values [1,R,1], normal_map [1,R,3], entropy [] of the last forward as views of ONE new
        allocation (7R + 1 floats), filled by one launch: the caller may keep them across iterations like the eager path's."""
        from .._native import CopySeg
        R, f = self.R, self.out["final"]
        srcs = (f["rgb_values"], f["depth_values"], f["normal_map"], f["entropy"])
        if not all(t.is_contiguous() and t.dtype == torch.float32 for t in srcs):
            return tuple(t.clone() for t in srcs)
        ptrs = tuple(t.data_ptr() for t in srcs)
        if self._out_segs is None or self._out_segs[0] != ptrs:      # (the eager warm-up call and the capture use other buffers)
            self._out_segs = (ptrs, (CopySeg * 4)(CopySeg(None, ptrs[0], 3 * R), CopySeg(None, ptrs[1], R),
                                                  CopySeg(None, ptrs[2], 3 * R), CopySeg(None, ptrs[3], 1)))
        segs = self._out_segs[1]
        buf = torch.empty(7 * R + 1, device=self.pose_s.device)
        base = buf.data_ptr()
        segs[0].dst, segs[1].dst, segs[2].dst, segs[3].dst = base, base + 12 * R, base + 16 * R, base + 28 * R
        check(lib.nsa_copy_segments(segs, 4, torch.cuda.current_stream().cuda_stream))
        return (buf[:3 * R].view(1, R, 3), buf[3 * R:4 * R].view(1, R, 1), buf[4 * R:7 * R].view(1, R, 3), buf[7 * R:].view(()))

    def backward(self, g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy):
        """Cotangents of (rgb_values, depth_values, normal_map, weights, entropy) -> d / d pose."""
        with torch.no_grad():
            only_rgb = g_rgbv is not None and g_depth_values is None and g_normal_map is None and g_w is None and g_entropy is None
            if only_rgb and self.bwd_graph is not None:
                self.g_rgbv_s.copy_(g_rgbv.reshape(self.R, 3))
                self.bwd_graph.replay()
                return self.g_pose
            if all(g is None for g in (g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy)):
                return None
            # any other objective: back through the output assembly with torch ops, then the backward kernels eagerly
            R, o = self.R, self.out
            g_depth = None if g_depth_values is None else (g_depth_values.reshape(R) * o["ds"]).contiguous()
            g_ent = None if g_entropy is None else (g_entropy / R).expand(R).contiguous()
            rot = self.pose_s[0, :3, :3]
            g_nmap = None if g_normal_map is None else torch.matmul(g_normal_map.reshape(R, 3), rot.t()).contiguous()
            g_rgb = None if g_rgbv is None else g_rgbv.reshape(R, 3).contiguous()
            g_pose = self._backward_body(g_rgb, g_depth, g_nmap, g_ent, None if g_w is None else g_w.contiguous())
            if g_normal_map is not None:          # normal_map = n @ R also depends on the pose directly
                g_pose = g_pose.clone()
                g_pose[0, :3, :3] += torch.matmul(o["b"]["nmap"].reshape(R, 3).t(), g_normal_map.reshape(R, 3))
            return g_pose


class _TrackingCore(torch.autograd.Function):
    """pose[1,4,4] (+ uv, K via the cache's static buffers) -> the tensors of SLAMNetwork.forward's output dict; backward to the pose."""

    @staticmethod
    def forward(ctx, pose, uv, K, tg):
        o = tg.forward(pose, uv, K)
        b, z, f = o["b"], o["z_vals"], o["final"]
        R, S = z.shape
        ctx.tg, ctx.serial = tg, tg.serial
        ctx.set_materialize_grads(False)
        sdf_o, rgb_o, w_o, dv_o = b["sdf"].view(R, S), b["rgb"].view(R, S, 3), b["weights"], f["depth_vals"]
        if CLONE == "none":
            rgbv, depthv, nmap, ent = f["rgb_values"], f["depth_values"], f["normal_map"], f["entropy"]
        else:
            rgbv, depthv, nmap, ent = tg.fresh_small_outputs()
            if CLONE == "all":
                sdf_o, rgb_o, w_o, z, dv_o = sdf_o.clone(), rgb_o.clone(), w_o.clone(), z.clone(), dv_o.clone()
        ctx.mark_non_differentiable(sdf_o, rgb_o, z, dv_o)
        return rgbv, depthv, nmap, w_o, ent, sdf_o, rgb_o, z, dv_o

    @staticmethod
    def backward(ctx, g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy, *_unused):
        tg = ctx.tg
        if ctx.serial != tg.serial:
            raise RuntimeError("SLAMNetwork tracking graph: backward through the outputs of an EARLIER forward -- the cached graph's "
                               "static buffers were overwritten by a later forward(mode='tracking'); call backward before the next "
                               "forward (the reference's loop does), or set NSA_TRACK_GRAPH=0")
        return tg.backward(g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy), None, None, None


def render(model, input, stage, color_stage):
    """-> the output dict of SLAMNetwork.forward(mode="tracking") from the cached graphs."""
    pose, uv, K = input["pose"], input["uv"], input["intrinsics"]
    R = uv.shape[1]
    key = _key(model, R, stage, color_stage)
    cache = model.__dict__.setdefault("_track_graphs", {})
    tg = cache.get("tg")
    if tg is None or cache.get("key") != key:
        cache.clear()                                  # (drops the old graphs and their pools)
        tg = TrackingGraph(model, R, stage, color_stage)
        cache["tg"], cache["key"] = tg, key
    rgb_values, depth_values, normal_map, weights, entropy, sdf, rgb, z_vals, depth_vals = _TrackingCore.apply(pose, uv, K, tg)
    return {"rgb": rgb, "rgb_values": rgb_values, "depth_values": depth_values, "z_vals": z_vals, "depth_vals": depth_vals,
            "sdf": sdf, "weights": weights, "entropy": entropy, "scene_bounding_sphere": model.scene_bounding_sphere,
            "normal_map": normal_map}
