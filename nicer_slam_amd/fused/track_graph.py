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

The tracking objective through the reference's own seams (round 6).  The reference's loop hands ``ground_truth`` to the model as well as
to the loss (volsdf_train.py:417-421), and its tracking objective is ``tracking_loss = SLAMLoss(rgb L1 only)`` resolved from the conf string
``train.loss_class`` (:117-130).  When ``ground_truth["rgb"]`` is a [1,R,3] tensor the cached forward graph therefore also runs the
objective the kernel tracker runs -- the ray's rendered colour, the L1 term and the composite backward in the colour forward's launch
(nsa_colour_forward_track; nsa_composite_track for sample counts other than 128) -- and the output dict carries
``out["tracking_rgb_l1"] = (loss, the ground-truth tensor it was formed against)``: a scalar whose backward replays a graph WITHOUT the
cotangent copy and the composite backward.  ``nicer_slam_amd.model.loss.SLAMLoss`` in its tracking configuration returns that scalar when
it is handed the same ground-truth tensor (no torch launch in the loss forward, one scaling launch in its backward); any other loss, or
any other use of the dict, runs exactly as before (the rgb-only backward graph stays captured beside it).

NSA_TRACK_GRAPH=0 keeps the eager Functions (A/B runs)."""
import os

import torch

from .._native import lib, check
from ..hashencoder.backend import _timed

ENABLED = os.environ.get("NSA_TRACK_GRAPH", "1") != "0"
CLONE = os.environ.get("NSA_TRACK_CLONE", "small")      # small | all | none (see the aliasing contract above)
OBJECTIVE = os.environ.get("NSA_TRACK_OBJECTIVE", "1") != "0"      # fold the tracking objective into the forward when ground_truth["rgb"] is given


# ---- packed MLP snapshots owned by a long-lived consumer (a captured graph reads their addresses) ---------------------------------
def pack_specs(model, n_rays, stage):
    """(cache key, network, use) of every packed block a tracking iteration over ``n_rays`` rays reads (fused/sampler.py::tile_of)."""
    from . import sampler as fs
    use = fs.sampler_use(n_rays)
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


def objective_gt(ground_truth, R):
    """ground_truth["rgb"] when the forward can fold the tracking objective in: a float32 [1,R,3] (or [R,3]) tensor; else None"""
    gt = ground_truth.get("rgb") if isinstance(ground_truth, dict) else None
    if torch.is_tensor(gt) and gt.dtype == torch.float32 and gt.numel() == 3 * R and gt.shape[-1] == 3 and not gt.requires_grad:
        return gt
    return None


class TrackingGraph:
    def __init__(self, model, R, stage, color_stage, objective=False):
        dev = model.voxels.device
        self.model, self.R, self.stage, self.color_stage = model, R, stage, color_stage
        self.objective = bool(objective)
        self.gt_s = torch.zeros(R, 3, device=dev) if objective else None
        self.ray_loss = torch.zeros(R, device=dev) if objective else None
        self.bwd_graph_obj, self.g_pose_obj = None, None
        self.pose_s = torch.zeros(1, 4, 4, device=dev)
        self.uv_s = torch.zeros(1, R, 2, device=dev)
        self.K = torch.zeros(1, 4, 4, device=dev)
        # every MLP parameter a packed block is built from: their version counters say when to re-pack (cheap per-call check)
        self._mlp_params = (list(model.implicit_network.coarse.mlp_parameters()) + list(model.implicit_network.fine.mlp_parameters())
                            + list(model.rendering_network.mlp_parameters()))
        self._mlp_stamp = None
        self.g_rgbv_s = torch.zeros(R, 3, device=dev)
        from .._native import CopySeg
        self._segs = (CopySeg * 4)(CopySeg(self.pose_s.data_ptr(), None, 16), CopySeg(self.uv_s.data_ptr(), None, 2 * R),
                                   CopySeg(self.K.data_ptr(), None, 16),
                                   CopySeg(self.gt_s.data_ptr() if objective else None, None, 3 * R))
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
        if fs.own_draws(model):
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
        if self.objective:
            # the tracking objective rides in the forward: rendered colour, per-ray L1 and the composite backward of d L1 / d rgb_values
            S = z_vals.shape[1]
            b = fr.composite_forward_raw(model, rays_o, rays_d, z_vals, self.stage, True, composite=False,
                                         track=dict(gt=self.gt_s, ray_loss=self.ray_loss))
            if "track_out" not in b:          # other sample counts than 128 per ray: the stand-alone per-ray launch
                P = R * S
                t_out = dict(g_sdf=torch.empty(P, device=dev), g_rgb=torch.empty(P, 3, device=dev), g_grad=torch.empty(P, 3, device=dev))
                with _timed("k_composite_track", P * 60):
                    check(lib.nsa_composite_track(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), b["sdf"].data_ptr(),
                                                  b["rgb"].data_ptr(), b["vox"].data_ptr(), model.voxel_res, R, S, self.gt_s.data_ptr(),
                                                  R, b["rgb_values"].data_ptr(), self.ray_loss.data_ptr(), t_out["g_sdf"].data_ptr(),
                                                  t_out["g_rgb"].data_ptr(), t_out["g_grad"].data_ptr(), st))
                b["track_out"] = t_out
            # ... and the rest of the output dict (weights, depth, normal map, entropy; the same rendered colour once more)
            with _timed("k_composite_fwd", R * S * 32):
                check(lib.nsa_composite_forward(rays_o.data_ptr(), rays_d.data_ptr(), z_vals.data_ptr(), b["sdf"].data_ptr(),
                                                b["rgb"].data_ptr(), b["grad"].data_ptr(), b["vox"].data_ptr(), model.voxel_res,
                                                R, S, b["weights"].data_ptr(), b["rgb_values"].data_ptr(), b["depth"].data_ptr(),
                                                b["nmap"].data_ptr(), b["entropy"].data_ptr(), st))
        else:
            b = fr.composite_forward_raw(model, rays_o, rays_d, z_vals, self.stage, True)
        # the forward's output dict (network.py:147-151, 281-300, 338-345) inside the same graph: five small torch launches that
        # would otherwise be dispatched, and recorded by autograd, on every call of the caller's loop
        rot = self.pose_s[:, :3, :3]
        final = dict(rgb_values=b["rgb_values"].view(1, R, 3), depth_values=(ds * b["depth"]).view(1, R, 1),
                     normal_map=torch.matmul(b["nmap"].view(1, R, 3), rot), entropy=b["entropy"].mean(),
                     depth_vals=z_vals * ds.view(R, 1))
        if self.objective:                   # mean |rgb_values - gt| over R x 3 (torch.nn.L1Loss, loss.py:57-65)
            final["objective"] = self.ray_loss.sum() / (3.0 * R)
        self.out = dict(b=b, rays_o=rays_o, rays_d=rays_d, ds=ds, z_vals=z_vals, z_eik=z_eik, final=final)

    def _backward_body(self, g_rgbv, g_depth=None, g_nmap=None, g_ent=None, g_w=None, objective=False):
        """objective=True: the cotangents of the folded tracking objective (unit upstream gradient) wait in b["track_out"]"""
        from . import render as fr
        o = self.out
        g_o, g_d = fr.composite_backward_raw(self.model, o["rays_o"], o["rays_d"], o["z_vals"], o["b"], self.stage, self.color_stage,
                                             g_rgbv, g_depth, g_nmap, g_ent, g_w,
                                             track=dict(gt=self.gt_s, ray_loss=self.ray_loss) if objective else None)
        g_pose = torch.empty(1, 4, 4, device=g_o.device)
        with _timed("k_rays_pose_bwd", self.R * 24):
            check(lib.nsa_rays_pose_backward(self.uv_s.data_ptr(), self.pose_s.data_ptr(), self.K.data_ptr(), 1, self.R,
                                             g_o.data_ptr(), g_d.data_ptr(), g_pose.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return g_pose

    def _copy_inputs(self, pose, uv, K, gt=None):
        """pose, uv, K (and the ground-truth colours of the folded objective) of this call into the graph's static buffers: ONE launch
        (nsa_copy_segments) when they are plain float32 device tensors, torch copies otherwise (strided / other dtype / host)."""
        R = self.R
        ok = lambda t, n: t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= n
        gt_seg = gt is not None and ok(gt, 3 * R) and gt.device == self.pose_s.device
        if gt is not None and not gt_seg:         # (the reference's own loader hands HOST tensors over; its loss uploads them, loss.py:121)
            self.gt_s.copy_(gt.detach().reshape(R, 3), non_blocking=True)
        if ok(pose, 16) and ok(uv, 2 * R) and ok(K, 16) and uv.numel() == 2 * R:
            segs = self._segs
            segs[0].src, segs[1].src, segs[2].src = pose.data_ptr(), uv.data_ptr(), K.data_ptr()
            if gt_seg:
                segs[3].src = gt.data_ptr()
            check(lib.nsa_copy_segments(segs, 4 if gt_seg else 3, torch.cuda.current_stream().cuda_stream))
        else:
            self.pose_s.copy_(pose.detach().reshape(1, 4, 4))
            self.uv_s.copy_(uv.detach())
            self.K.copy_(K.detach().reshape(-1, 4, 4)[:1])
            if gt_seg:
                self.gt_s.copy_(gt.detach().reshape(R, 3))

    def forward(self, pose, uv, K, gt=None):
        with torch.no_grad():
            self._copy_inputs(pose, uv, K, gt if self.objective else None)
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
                if self.objective:
                    self._backward_body(None, objective=True)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward_body()
                self.fwd_graph = g
                gb = torch.cuda.CUDAGraph()                         # the backward of the tracking objective reads the forward's
                with torch.cuda.graph(gb):                          # static buffers: capture it now, on this thread
                    self.g_pose = self._backward_body(self.g_rgbv_s)
                self.bwd_graph = gb
                if self.objective:                                  # ... and the one of the folded objective (no cotangent to copy in)
                    go = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(go):
                        self.g_pose_obj = self._backward_body(None, objective=True)
                    self.bwd_graph_obj = go
                g.replay()
        self.serial += 1
        return self.out

    def fresh_small_outputs(self):
        """rgb_values [1,R,3], depth_values [1,R,1], normal_map [1,R,3], entropy [] of the last forward as views of ONE new
        allocation (7R + 1 floats), filled by one launch: the caller may keep them across iterations like the eager path's."""
        from .._native import CopySeg
        R, f = self.R, self.out["final"]
        srcs = (f["rgb_values"], f["depth_values"], f["normal_map"], f["entropy"]) + ((f["objective"],) if self.objective else ())
        if not all(t.is_contiguous() and t.dtype == torch.float32 for t in srcs):
            return tuple(t.clone() for t in srcs) + (() if self.objective else (None,))
        ptrs = tuple(t.data_ptr() for t in srcs)
        n = len(srcs)
        if self._out_segs is None or self._out_segs[0] != ptrs:      # (the eager warm-up call and the capture use other buffers)
            sizes = (3 * R, R, 3 * R, 1, 1)
            self._out_segs = (ptrs, (CopySeg * n)(*[CopySeg(None, ptrs[i], sizes[i]) for i in range(n)]))
        segs = self._out_segs[1]
        buf = torch.empty(7 * R + 2, device=self.pose_s.device)
        base = buf.data_ptr()
        segs[0].dst, segs[1].dst, segs[2].dst, segs[3].dst = base, base + 12 * R, base + 16 * R, base + 28 * R
        if self.objective:
            segs[4].dst = base + 28 * R + 4
        check(lib.nsa_copy_segments(segs, n, torch.cuda.current_stream().cuda_stream))
        return (buf[:3 * R].view(1, R, 3), buf[3 * R:4 * R].view(1, R, 1), buf[4 * R:7 * R].view(1, R, 3), buf[7 * R:7 * R + 1].view(()),
                buf[7 * R + 1:].view(()) if self.objective else None)

    def backward(self, g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy, g_obj=None):
        """Cotangents of (rgb_values, depth_values, normal_map, weights, entropy[, the folded objective]) -> d / d pose."""
        with torch.no_grad():
            if g_obj is not None:
                if all(g is None for g in (g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy)) and self.bwd_graph_obj is not None:
                    self.bwd_graph_obj.replay()               # cotangents of the unit objective were formed in the forward
                    return self.g_pose_obj * g_obj
                # the objective beside other uses of the outputs, or before the graphs exist: its cotangent joins rgb_values'
                d = self.out["final"]["rgb_values"].reshape(self.R, 3) - self.gt_s
                g_l1 = torch.sign(d) * (g_obj / (3.0 * self.R))
                g_rgbv = g_l1 if g_rgbv is None else g_rgbv.reshape(self.R, 3) + g_l1
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
    def forward(ctx, pose, uv, K, tg, gt=None):
        o = tg.forward(pose, uv, K, gt)
        b, z, f = o["b"], o["z_vals"], o["final"]
        R, S = z.shape
        ctx.tg, ctx.serial = tg, tg.serial
        ctx.set_materialize_grads(False)
        sdf_o, rgb_o, w_o, dv_o = b["sdf"].view(R, S), b["rgb"].view(R, S, 3), b["weights"], f["depth_vals"]
        if CLONE == "none":
            rgbv, depthv, nmap, ent = f["rgb_values"], f["depth_values"], f["normal_map"], f["entropy"]
            obj = f["objective"] if tg.objective else None
        else:
            rgbv, depthv, nmap, ent, obj = tg.fresh_small_outputs()
            if CLONE == "all":
                sdf_o, rgb_o, w_o, z, dv_o = sdf_o.clone(), rgb_o.clone(), w_o.clone(), z.clone(), dv_o.clone()
        ctx.mark_non_differentiable(sdf_o, rgb_o, z, dv_o)
        if obj is None:
            obj = tg.__dict__.get("_no_obj")
            if obj is None:
                obj = tg._no_obj = rgbv.new_zeros(())
            ctx.mark_non_differentiable(obj)
        return rgbv, depthv, nmap, w_o, ent, sdf_o, rgb_o, z, dv_o, obj

    @staticmethod
    def backward(ctx, g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy, _g_sdf, _g_rgb, _g_z, _g_dv, g_obj):
        tg = ctx.tg
        if ctx.serial != tg.serial:
            raise RuntimeError("SLAMNetwork tracking graph: backward through the outputs of an EARLIER forward -- the cached graph's "
                               "static buffers were overwritten by a later forward(mode='tracking'); call backward before the next "
                               "forward (the reference's loop does), or set NSA_TRACK_GRAPH=0")
        return tg.backward(g_rgbv, g_depth_values, g_normal_map, g_w, g_entropy, g_obj if tg.objective else None), None, None, None, None


def render(model, input, stage, color_stage, ground_truth=None):
    """-> the output dict of SLAMNetwork.forward(mode="tracking") from the cached graphs."""
    pose, uv, K = input["pose"], input["uv"], input["intrinsics"]
    R = uv.shape[1]
    gt = objective_gt(ground_truth, R) if OBJECTIVE else None
    key = _key(model, R, stage, color_stage) + (gt is not None,)
    cache = model.__dict__.setdefault("_track_graphs", {})
    tg = cache.get("tg")
    if tg is None or cache.get("key") != key:
        cache.clear()                                  # (drops the old graphs and their pools)
        tg = TrackingGraph(model, R, stage, color_stage, objective=gt is not None)
        cache["tg"], cache["key"] = tg, key
    rgb_values, depth_values, normal_map, weights, entropy, sdf, rgb, z_vals, depth_vals, obj = _TrackingCore.apply(pose, uv, K, tg, gt)
    out = {"rgb": rgb, "rgb_values": rgb_values, "depth_values": depth_values, "z_vals": z_vals, "depth_vals": depth_vals,
           "sdf": sdf, "weights": weights, "entropy": entropy, "scene_bounding_sphere": model.scene_bounding_sphere,
           "normal_map": normal_map}
    if gt is not None:
        # mean |rgb_values - ground_truth["rgb"]| with its own backward, and the tensor it was formed against (model/loss.py uses it only
        # when it is handed that very tensor)
        out["tracking_rgb_l1"] = (obj, gt)
    return out
