"""One camera-tracking iteration as the reference's loop runs it (code/training/volsdf_train.py:406-443):
camera 7-vector -> c2w -> SLAMNetwork.forward(mode="tracking") -> L1(rgb) -> backward -> Adam step on the camera --
optionally captured into a hipGraph (torch.cuda.CUDAGraph), so that the ~100 small launches of the pose math, the loss,
autograd bookkeeping and Adam cost one graph launch instead of ~1 ms of host time.

The fused kernels are launched through the C ABI on torch's current stream, so they are captured like any torch op; the
random draws use the default device generator (philox offsets advance per replay)."""
import torch

from .dist import allreduce_pose_grad
from .hashencoder.backend import _timed
from .utils.general import get_camera_from_tensor


class KernelTracker:
    """The same iteration with NO autograd in the loop: a fixed sequence of our kernels launched through the C ABI, optionally one
    hipGraph.  One ray chunk (the default): nsa_track_begin_draw (batch copy, cam->pose, rays and the sampler's draws; in front of
    the graph) | nsa_sampler_sdf, nsa_sample_rays, nsa_sdfnet_forward_pair, nsa_colour_forward_track (colour forward, then the ray's composite +
    L1 + composite-bwd), nsa_colour_coarse_backward (colour + coarse SDF backward), nsa_sdfnet_backward (fine), nsa_track_finish
    (ray sums, pose-bwd, cam-bwd, Adam, candidate) -- eight launches; with ray chunks the plain entry points (head, composite, L1, composite-bwd, ray reduction, tail),
    and with N > 1 ranks [all-reduce] + nsa_adam_step_scaled after the message.
    Only tracking (pose gradient) is covered; it needs a configuration in the fused engine's compiled set.

    Per-frame protocol of the reference's tracking loop (volsdf_train.py:393-446): ``reset(cam_init)`` starts a frame
    (fresh Adam state, StepLR back to its first step, no candidate yet), ``step(uv, gt)`` runs one iteration, and
    ``candidate`` is the camera of the smallest loss seen -- cloned AFTER that iteration's optimizer step, like the
    reference -- which is what the frame's pose estimate is set to (:445-446).  ``lr_step=50, lr_gamma=0.95`` is the
    reference's StepLR (:398).

    ``chunks > 1`` splits the ray batch into independent contiguous chunks that run the whole kernel sequence on their own
    HIP streams (fork / join inside the captured graph) and meet only in the 9-float message [w*g_cam, w*loss, w] that the
    multi-GPU form already uses: rays are independent until the pose-gradient sum, and kernels of different character
    (VALU-bound SDF networks, gather-bound colour grid, latency-bound per-ray scans) then share the CUs and fill each
    other's tails.  Same arithmetic per ray; the ray sums are added chunk by chunk."""

    def __init__(self, model, intrinsics, n_rays, cam_init, lr=0.005, betas=(0.9, 0.999), eps=1e-8, lr_step=0,
                 lr_gamma=1.0, use_graph=True, world=1, stage="fine", color_stage="highfreq", chunks=1, graph_collective=None,
                 fold=True, draw_in_begin=True):
        from .fused import render as fr, sampler as fs
        if not fr.supported(model):
            raise RuntimeError("KernelTracker: configuration outside the fused engine's compiled set")
        dev = model.voxels.device
        self.fr, self.fs = fr, fs
        self.model, self.world, self.R = model, world, n_rays
        self.K = intrinsics.contiguous()
        self.stage, self.color_stage = stage, color_stage
        self.hyper = (float(lr), float(betas[0]), float(betas[1]), float(eps), int(lr_step), float(lr_gamma))
        self.cam = cam_init.detach().clone().to(dev).float().contiguous()
        self.uv = torch.zeros(1, n_rays, 2, device=dev)
        self.gt = torch.zeros(n_rays, 3, device=dev)
        self._uv_shape, self._gt_shape = (1, n_rays, 2), (n_rays, 3)
        z = lambda *s: torch.zeros(*s, device=dev)
        self.pose, self.g_pose, self.red = z(1, 4, 4), z(1, 4, 4), z(9)     # red = [g_cam(7), loss, n_rays]
        self.m, self.v, self.t = z(7), z(7), z(1)
        self.best = z(8)                                                    # [min loss, camera after that step (7)]
        self.best[0] = 1e10                                                 # current_min_loss (volsdf_train.py:403)
        self.rays_o, self.rays_d, self.ds = z(n_rays, 3), z(n_rays, 3), z(n_rays)
        self.g_rgbv = z(n_rays, 3)
        # multi-GPU: capture the 9-float all-reduce and the Adam step into the graph as well (RCCL collectives can be captured;
        # every rank captures the same sequence).  Opt-in (NSA_GRAPH_COLLECTIVE=1 or graph_collective=True): the build boxes have
        # one GPU, so only the 1-rank form of the captured collective could be exercised (tests/test_fused_gpu.py).
        import os
        if graph_collective is None:
            graph_collective = os.environ.get("NSA_GRAPH_COLLECTIVE", "0") == "1"
        self.collective_in_graph = bool(graph_collective and use_graph and world > 1)
        self.chunks = max(1, min(int(chunks), n_rays))
        if self.chunks > 1:
            from .dist import shard_rays
            self.bounds = [shard_rays(n_rays, c, self.chunks) for c in range(self.chunks)]
            self.streams = [torch.cuda.Stream() for _ in range(self.chunks)]
            self.red_c, self.pose_c = z(self.chunks, 9), z(self.chunks, 4, 4)
        # one ray chunk: the per-ray neighbours of head and tail are folded into them (nsa_track_begin in front of the graph,
        # nsa_composite_track, nsa_track_finish): 5 fewer launches per iteration.  fold=False keeps the plain sequence (the
        # bit-identity test of the folded forms, tests/test_track_fold_gpu.py).
        self.folded = self.chunks == 1 and bool(fold)
        if self.folded:
            from ._native import lib
            self.ray_loss = z(n_rays)
            self.fin_ws = z(int(lib.nsa_track_finish_workspace(n_rays)))      # ticket + block partials; zero once
        # folded sequence: the sampler's draws of an iteration are made by the head launch as well (nsa_track_begin_draw) instead of
        # a graph node of their own -- same generator state, same numbers.  draw_in_begin=False: nsa_draw inside the graph (same test).
        self.drawn = None
        if self.folded and draw_in_begin:
            samp = model.ray_sampler
            E, n_extra = samp.N_samples_eval, samp.N_samples_extra
            if E <= 1024:
                self.drawn = (torch.empty(n_rays, E, device=dev),
                              torch.empty(max(n_extra, 1), device=dev, dtype=torch.int32)[:n_extra] if n_extra > 0 else None)
        self._drawn_now = False    # whether the coming iteration's draws were made by _begin
        self.graph = None
        self._began = False        # folded sequence: _begin (batch copy + cam -> pose -> rays) has run for the coming iteration
        if use_graph:
            self._capture()

    @property
    def message(self):
        """the step consumes the weighted 9-float message (multi-GPU and / or chunked) instead of the plain gradient"""
        return self.world > 1 or self.chunks > 1

    @property
    def loss(self):
        return self.red[7] / self.red[8] if self.message else self.red[7]

    @property
    def candidate(self):
        """camera 7-vector of the smallest loss of this frame (candidate_cam_tensor, volsdf_train.py:441-446)"""
        return self.best[1:8]

    @property
    def min_loss(self):
        return self.best[0]

    def reset(self, cam_init):
        """Start a new frame: camera estimate, Adam moments / step counter (a new optimizer + StepLR per frame,
        volsdf_train.py:396-398) and the candidate -- all in place, so a captured graph keeps replaying on them."""
        with torch.no_grad():
            self.cam.copy_(cam_init.detach().to(self.cam.device).float())
            for t in (self.m, self.v, self.t):
                t.zero_()
            self.best.zero_()
            self.best[0] = 1e10

    def _rays_pass(self, lo, hi, pose, red):
        """head .. tail for the rays [lo, hi) on torch's current stream; `red` receives g_cam (and, fused, the Adam step) or the
        weighted message"""
        from ._native import lib, check
        fr, fs, model = self.fr, self.fs, self.model
        st = torch.cuda.current_stream().cuda_stream
        R = hi - lo
        uv, gt, g_rgbv = self.uv[0, lo:hi], self.gt[lo:hi], self.g_rgbv[lo:hi]
        rays_o, rays_d, ds = self.rays_o[lo:hi], self.rays_d[lo:hi], self.ds[lo:hi]
        lr, b1, b2, eps, lr_step, lr_gamma = self.hyper
        fused = not self.message
        if self.folded:                                # (the rays were lifted by _begin, in front of the graph)
            if not self._began and not torch.cuda.is_current_stream_capturing():
                raise RuntimeError("KernelTracker: the folded sequence needs _begin() (nsa_track_begin) in front of every iteration -- "
                                   "the rays of the updated camera are lifted there, not inside the graph; use step()")
            self._began = False
            z_vals, _ = fs.get_z_vals(model, rays_d, rays_o, need_eik=False, rows=(lo, hi),
                                      drawn=self.drawn if self._drawn_now else None)
            trk = dict(gt=gt, ray_loss=self.ray_loss)
            b = fr.composite_forward_raw(model, rays_o, rays_d, z_vals, self.stage, True, composite=False, track=trk)
            g_x, g_dir = fr.composite_backward_raw(model, rays_o, rays_d, z_vals, b, self.stage, self.color_stage,
                                                   track=trk, reduce_rays=False)
            with _timed("k_track_finish", R * z_vals.shape[1] * 24):
                check(lib.nsa_track_finish(uv.data_ptr(), self.K.data_ptr(), self.cam.data_ptr(), R, z_vals.shape[1],
                                           z_vals.data_ptr(), g_x.data_ptr(), g_dir.data_ptr(), self.ray_loss.data_ptr(),
                                           red.data_ptr(), 1 if fused else 0, 0.0 if fused else float(R), self.m.data_ptr(),
                                           self.v.data_ptr(), self.t.data_ptr(), lr, b1, b2, eps, lr_step, lr_gamma,
                                           self.best.data_ptr() if fused else None, self.fin_ws.data_ptr(), st))
            return
        check(lib.nsa_track_head(uv.data_ptr(), self.K.data_ptr(), self.cam.data_ptr(), R, pose.data_ptr(),
                                 rays_o.data_ptr(), rays_d.data_ptr(), ds.data_ptr(), st))
        z_vals, _ = fs.get_z_vals(model, rays_d, rays_o, need_eik=False, rows=(lo, hi))
        b = fr.composite_forward_raw(model, rays_o, rays_d, z_vals, self.stage, True)
        check(lib.nsa_l1_loss(b["rgb_values"].data_ptr(), gt.data_ptr(), 3 * R, red[7:8].data_ptr(), g_rgbv.data_ptr(), st))
        g_o, g_d = fr.composite_backward_raw(model, rays_o, rays_d, z_vals, b, self.stage, self.color_stage, g_rgbv=g_rgbv)
        # one chunk on one GPU: the Adam step rides in the same kernel; otherwise the tail leaves the weighted message and the
        # step follows the chunk sum / the all-reduce
        check(lib.nsa_track_tail(uv.data_ptr(), self.K.data_ptr(), self.cam.data_ptr(), R, g_o.data_ptr(),
                                 g_d.data_ptr(), red.data_ptr(), 1 if fused else 0, 0.0 if fused else float(R),
                                 self.m.data_ptr(), self.v.data_ptr(), self.t.data_ptr(), lr, b1, b2, eps, lr_step, lr_gamma,
                                 red[7:8].data_ptr() if fused else None, self.best.data_ptr() if fused else None, st))

    # ---- packed MLP parameters (fused/pack.py) ---------------------------------------------------------------------------
    # The kernels read the weight-normed MLP parameters from packed snapshots cached on (data_ptr, _version) of the parameters.
    # A captured graph holds the snapshots' ADDRESSES: when a mapping step between two tracked frames changes the MLPs, the
    # next cache miss would allocate a new snapshot (and free the captured one) while the graph kept replaying on the stale --
    # possibly recycled -- memory.  The tracker therefore owns the snapshots its sequence reads: before every iteration it
    # compares the parameters' keys with those of its snapshots and, on a change, re-packs INTO the same tensors.
    def _pack_specs(self):
        from .fused.track_graph import pack_specs
        return pack_specs(self.model, self.R // self.chunks, self.stage)

    def _ensure_packs(self):
        """Fresh packed blocks for every tiling the sequence uses, on the current stream (so forked chunk streams never race a
        cache miss), re-packed in place when the tracker already owns them (fused/track_graph.py::ensure_packs)."""
        from .fused.track_graph import ensure_packs
        ensure_packs(self.model, self.__dict__.setdefault("_packs", {}), self._pack_specs())

    def _iteration(self):
        if self.graph is None or not torch.cuda.is_current_stream_capturing():
            self._ensure_packs()
        if self.chunks == 1:
            self._rays_pass(0, self.R, self.pose, self.red)
            if self.collective_in_graph:
                self._exchange()
            return
        cur = torch.cuda.current_stream()
        for c, side in enumerate(self.streams):          # fork
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self._rays_pass(*self.bounds[c], self.pose_c[c], self.red_c[c])
        for side in self.streams:                        # join
            cur.wait_stream(side)
        torch.sum(self.red_c, 0, out=self.red)
        if self.world == 1:
            self._update()
        elif self.collective_in_graph:
            self._exchange()

    def _exchange(self):
        """weighted mean over the global ray batch: the tail kernel(s) left the 9-float message"""
        import torch.distributed as dist
        dist.all_reduce(self.red)
        self._update()

    def _update(self):
        """Adam on the (chunk-summed, all-reduced) message, gradient = red[0..6] / red[8]"""
        from ._native import lib, check
        lr, b1, b2, eps, lr_step, lr_gamma = self.hyper
        check(lib.nsa_adam_step_scaled(self.cam.data_ptr(), self.red.data_ptr(), self.red[8:].data_ptr(), self.m.data_ptr(),
                                       self.v.data_ptr(), self.t.data_ptr(), 7, lr, b1, b2, eps, lr_step, lr_gamma,
                                       self.red[7:8].data_ptr(), self.best.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream))

    def _begin(self, uv, gt):
        """The frame's pixel batch into the resident buffers + camera -> pose -> rays, one launch (folded sequence)."""
        from ._native import lib, check
        dev = self.uv.device
        uv = uv.detach().to(device=dev, dtype=torch.float32).reshape(-1, 2).contiguous()
        gt = gt.detach().to(device=dev, dtype=torch.float32).reshape(-1, 3).contiguous()
        if uv.shape[0] != self.R or gt.shape[0] != self.R:
            raise ValueError(f"KernelTracker.step: expected {self.R} rays, got uv {tuple(uv.shape)} / gt {tuple(gt.shape)}")
        # a captured graph replays the choice made at capture; an eager tracker follows the model's state (pinned draws, eval mode)
        draw = (self.drawn is not None and self.fs.own_draws(self.model)) if self.graph is None else self._drawn_now
        st = torch.cuda.current_stream().cuda_stream
        with _timed("k_track_begin", self.R * 64):
            if draw:
                samp = self.model.ray_sampler
                t_rand, extra = self.drawn
                check(lib.nsa_track_begin_draw(uv.data_ptr(), gt.data_ptr(), self.uv.data_ptr(), self.gt.data_ptr(), self.K.data_ptr(),
                                               self.cam.data_ptr(), self.R, self.pose.data_ptr(), self.rays_o.data_ptr(),
                                               self.rays_d.data_ptr(), self.ds.data_ptr(), self.fs.draw_state(self.model, 0).data_ptr(),
                                               t_rand.numel(), t_rand.data_ptr(), samp.N_samples_eval, samp.N_samples_extra,
                                               samp.N_samples + 2 + samp.N_samples_extra,
                                               None if extra is None else extra.data_ptr(), st))
            else:
                check(lib.nsa_track_begin(uv.data_ptr(), gt.data_ptr(), self.uv.data_ptr(), self.gt.data_ptr(), self.K.data_ptr(),
                                          self.cam.data_ptr(), self.R, self.pose.data_ptr(), self.rays_o.data_ptr(),
                                          self.rays_d.data_ptr(), self.ds.data_ptr(), st))
        if self.graph is None:
            self._drawn_now = draw
        self._began = True

    def _capture(self):
        cam0 = self.cam.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                if self.folded:
                    self._begin(self.uv, self.gt)
                self._iteration()
        torch.cuda.current_stream().wait_stream(side)
        self.reset(cam0)                           # the warm-up iterations stepped the camera: undo
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._iteration()
        self.reset(cam0)                           # (capture does not execute, but keep the state explicit)

    def step(self, uv, gt):
        if self.folded:
            self._begin(uv, gt)
        else:
            self.uv.copy_(uv)
            self.gt.copy_(gt)
        with torch.no_grad():
            if self.graph is not None:
                self._ensure_packs()               # the graph reads the tracker-owned snapshots: refresh them in place
                self.graph.replay()
                self._began = False
            else:
                self._iteration()
            if self.world > 1 and not self.collective_in_graph:
                self._exchange()
        return self.loss


class TrackingStepper:
    """The iteration driven through torch autograd exactly as volsdf_train.py:406-446 does: get_camera_from_tensor ->
    SLAMNetwork.forward(mode="tracking") -> L1 -> loss.backward() -> torch.optim.Adam (+ StepLR) -> arg-min-loss candidate."""

    def __init__(self, model, intrinsics, n_rays, cam_init, lr=0.005, use_graph=True, world=1, lr_step=0, lr_gamma=1.0,
                 opt_cls=None, loss_fn=None):
        """loss_fn: the tracking objective as the reference's loop calls it -- ``loss_fn(model_outputs, ground_truth, stage="fine",
        frame_idx=...)["loss"]`` with the `tracking_loss` SLAMLoss instance (volsdf_train.py:117-130, 418-421); the frame's ground truth is
        then handed to the model as well, as that loop does (:417).  None: the L1 term written out inline."""
        dev = model.voxels.device
        self.loss_fn = loss_fn
        self.model, self.world, self.n_rays = model, world, n_rays
        self.K = intrinsics
        self.cam = cam_init.detach().clone().to(dev).requires_grad_(True)
        self.uv = torch.zeros(1, n_rays, 2, device=dev)
        self.gt = torch.zeros(n_rays, 3, device=dev)
        self._uv_shape, self._gt_shape = (1, n_rays, 2), (n_rays, 3)
        self.ind = torch.zeros(1, dtype=torch.long, device=dev)
        self.graph_all = use_graph and world == 1     # Adam inside the graph only when no all-reduce sits in between
        # opt_cls: a drop-in replacement for torch.optim.Adam with the same constructor (nicer_slam_amd.optim.Adam: one launch)
        self.opt = (torch.optim.Adam([self.cam], lr=lr, capturable=self.graph_all) if opt_cls is None
                    else opt_cls([self.cam], lr=lr))
        if lr_step and self.graph_all:
            raise ValueError("TrackingStepper: StepLR is stepped on the host -- use use_graph=False (or KernelTracker, whose "
                             "Adam kernel applies the schedule on the device)")
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, lr_step, lr_gamma) if lr_step else None
        self.min_loss = torch.full((), 1e10, device=dev)           # current_min_loss / candidate_cam_tensor (:402-403)
        self.candidate = self.cam.detach().clone()
        self.graph = None
        self.loss = None
        if use_graph:
            self._capture()

    def _fwd_bwd(self):
        pose = get_camera_from_tensor(self.cam).unsqueeze(0)
        if self.loss_fn is not None:       # volsdf_train.py:415-424: the loader's ground-truth dict goes to the model AND to the loss
            gt = self._ground_truth()
            out = self.model({"intrinsics": self.K, "uv": self.uv, "pose": pose}, self.ind, gt, mode="tracking", frame_idx=1)
            loss = self.loss_fn(out, gt, stage="fine", frame_idx=1)["loss"]
        else:
            out = self.model({"intrinsics": self.K, "uv": self.uv, "pose": pose}, self.ind, {}, mode="tracking", frame_idx=1)
            loss = (out["rgb_values"].reshape(-1, 3) - self.gt).abs().mean()     # SLAMLoss.get_rgb_loss, L1Loss(mean)
        loss.backward()
        return loss.detach()

    def _ground_truth(self):
        """the dict a data loader's collate hands over for one tracked frame (scene_dataset.py:262-287): rgb + the cue tensors the
        tracking objective gives zero weight"""
        R = self.n_rays
        cues = self.__dict__.get("_cues")
        if cues is None:
            z = lambda c: torch.zeros(1, R, c, device=self.cam.device)
            cues = self._cues = {"depth": z(1), "normal": z(3), "gt_depth": z(1), "mask": torch.ones(1, R, 1, device=self.cam.device)}
        return dict(cues, rgb=self.gt.view(1, R, 3))

    def _eager(self):
        self.opt.zero_grad(set_to_none=False) if self.cam.grad is not None else None
        loss = self._fwd_bwd()
        return loss

    def _capture(self):
        cam0 = self.cam.detach().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):                      # warm-up: allocator pools, packed weights, host offsets, lazy init
                self.cam.grad = None
                self._fwd_bwd()
                if self.graph_all:
                    self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        # undo the warm-up: same camera and a fresh optimizer state (zeroed IN PLACE: the graph captures these tensors)
        with torch.no_grad():
            self.cam.copy_(cam0)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        self.cam.grad = None
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd()
            if self.graph_all:
                self.opt.step()

    def step(self, uv, gt):
        """uv [1,R,2] float pixels, gt [R,3] colours (device tensors).  Returns the (device) loss of this iteration."""
        if self.graph is not None:
            self.uv.copy_(uv)
            self.gt.copy_(gt)
            self.graph.replay()
            loss = self.loss
        else:
            # eager: no static inputs to fill (the reference's loop has none either) -- but only a float32 device tensor of the
            # expected shape is used as is; anything else (host tensor, other dtype, [R,2] pixels) is normalised by a copy
            dev = self.cam.device
            def _as(t, shape):
                if t.is_cuda and t.device == dev and t.dtype == torch.float32 and tuple(t.shape) == shape and t.is_contiguous():
                    return t
                return t.detach().to(device=dev, dtype=torch.float32).reshape(shape).contiguous()
            self.uv, self.gt = _as(uv, tuple(self._uv_shape)), _as(gt, tuple(self._gt_shape))
            # optimizer_camera.zero_grad() of the reference's loop (volsdf_train.py:427): under torch >= 2.0 that sets .grad = None
            # (a zero fill + an accumulating add less per iteration than zeroing in place)
            self.opt.zero_grad()
            loss = self._fwd_bwd()
        if not self.graph_all:
            if self.world > 1:
                g, loss = allreduce_pose_grad(self.cam.grad, loss, self.n_rays)
                self.cam.grad.copy_(g)
            self.opt.step()
        if self.sched is not None:
            self.sched.step()
        with torch.no_grad():                                       # if loss < current_min_loss: clone the (stepped) camera
            better = loss < self.min_loss
            self.min_loss = torch.where(better, loss, self.min_loss)
            self.candidate = torch.where(better, self.cam.detach(), self.candidate)
        return loss
