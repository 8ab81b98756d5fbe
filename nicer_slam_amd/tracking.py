"""One camera-tracking iteration as the reference's loop runs it (code/training/volsdf_train.py:406-443):
camera 7-vector -> c2w -> SLAMNetwork.forward(mode="tracking") -> L1(rgb) -> backward -> Adam step on the camera --
optionally captured into a hipGraph (torch.cuda.CUDAGraph), so that the ~100 small launches of the pose math, the loss,
autograd bookkeeping and Adam cost one graph launch instead of ~1 ms of host time.

The fused kernels are launched through the C ABI on torch's current stream, so they are captured like any torch op; the
random draws use the default device generator (philox offsets advance per replay)."""
import torch

from .dist import allreduce_pose_grad
from .utils.general import get_camera_from_tensor


class TrackingStepper:
    def __init__(self, model, intrinsics, n_rays, cam_init, lr=0.005, use_graph=True, world=1):
        dev = model.voxels.device
        self.model, self.world, self.n_rays = model, world, n_rays
        self.K = intrinsics
        self.cam = cam_init.detach().clone().to(dev).requires_grad_(True)
        self.uv = torch.zeros(1, n_rays, 2, device=dev)
        self.gt = torch.zeros(n_rays, 3, device=dev)
        self.ind = torch.zeros(1, dtype=torch.long, device=dev)
        self.graph_all = use_graph and world == 1     # Adam inside the graph only when no all-reduce sits in between
        self.opt = torch.optim.Adam([self.cam], lr=lr, capturable=self.graph_all)
        self.graph = None
        self.loss = None
        if use_graph:
            self._capture()

    def _fwd_bwd(self):
        pose = get_camera_from_tensor(self.cam).unsqueeze(0)
        out = self.model({"intrinsics": self.K, "uv": self.uv, "pose": pose}, self.ind, {}, mode="tracking", frame_idx=1)
        loss = (out["rgb_values"].reshape(-1, 3) - self.gt).abs().mean()     # SLAMLoss.get_rgb_loss, L1Loss(mean)
        loss.backward()
        return loss.detach()

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
        self.uv.copy_(uv)
        self.gt.copy_(gt)
        if self.graph is not None:
            self.graph.replay()
            loss = self.loss
        else:
            if self.cam.grad is not None:
                self.cam.grad.zero_()
            loss = self._fwd_bwd()
        if not self.graph_all:
            if self.world > 1:
                g, loss = allreduce_pose_grad(self.cam.grad, loss, self.n_rays)
                self.cam.grad.copy_(g)
            self.opt.step()
        return loss
