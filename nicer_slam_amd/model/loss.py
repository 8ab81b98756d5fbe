"""Mapping / tracking objective: the per-ray loss terms of the reference's SLAMLoss, restated.

Reference: SLAMLoss (code/model/loss.py:8-233) and the scale-and-shift-invariant monocular depth loss it uses
(code/utils/MiDaS.py:6-143, alpha = 0.5, one scale, batch-based reduction).  Same constructor arguments, same
``forward(model_outputs, ground_truth, keyframe_list, frame_idx, stage) -> dict`` keys and weights, so a trainer that
builds ``loss_class(**conf.loss, ...)`` can switch.  On the GPU the per-ray terms and their gradients run in three HIP launches
(fused/loss.py -> csrc/loss_terms.hip: rgb, eikonal, smooth, ssi depth, gt depth, normals); the torch restatement below is the
same arithmetic for CPU tensors, for the ray-sharded depth reduction and for non-L1 colour losses; the flow and patch-warp terms
are masked-L1 means (fused/warp.py::masked_l1 -> csrc/warp_terms.hip on the GPU) over the tensors the model's re-projection
kernels produce.  The tracking objective (rgb L1 only) additionally exists as a HIP kernel for the
graph-captured tracker (csrc/track_tail.hip::k_l1_loss).
"""
import torch
from torch import nn

from ..utils.general import get_class


def _masked_mean_abs(a, b, mask=None):
    if mask is not None:
        m = mask.reshape(-1)
        a, b = a[m], b[m]
    return (a - b).abs().mean()


def scale_shift_invariant_depth_loss(pred, target, mask, alpha=0.5, shard=None):
    """Least-squares (scale, shift) per image aligning ``pred`` to ``target`` on ``mask`` (closed-form 2x2 solve,
    detached), then  sum(mask (s p + t - target)^2) / (2 sum mask)  +  alpha * masked first-difference L1 of the residual
    along both pixel axes / sum mask.  Shapes [b, n, 1] (the reference feeds rays as an n x 1 'image', so only the
    difference along n is non-empty).  MiDaS.py:6-143.

    ``shard`` = (group, weight): ray-sharded multi-GPU mapping.  Every rank holds a slice of each keyframe's
    rays; the five per-image sums of the 2x2 system and the two normalisers are summed over ranks with ONE all-reduce, so every
    rank solves the same (scale, shift) as a single process would; the returned value is this rank's share of the terms divided
    by ``weight`` (its share of the ray batch), so that  sum_r weight_r * loss_r  -- what ShardedAdam forms from the ranks'
    gradients -- is the single-process loss.  (The first-difference regulariser is taken inside each rank's slice: rays are
    sharded contiguously, the one pair straddling a shard boundary is dropped.)"""
    mask = mask.to(pred.dtype)
    dims = (1, 2)
    a00 = (mask * pred * pred).sum(dims)
    a01 = (mask * pred).sum(dims)
    a11 = mask.sum(dims)
    b0 = (mask * pred * target).sum(dims)
    b1 = (mask * target).sum(dims)
    weight = 1.0
    if shard is not None:
        import torch.distributed as dist
        group, weight = shard
        sums = torch.stack([a00, a01, a11, b0, b1]).detach()          # [5, b]: the solve is detached anyway (MiDaS.py:22-26)
        dist.all_reduce(sums, group=group)
        a00, a01, a11, b0, b1 = sums.unbind(0)
    det = a00 * a11 - a01 * a01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    scale = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.zeros_like(det)).detach()
    shift = torch.where(ok, (a00 * b1 - a01 * b0) / safe, torch.zeros_like(det)).detach()
    aligned = scale.view(-1, 1, 1) * pred + shift.view(-1, 1, 1)
    M = a11 if shard is not None else mask.sum(dims)                   # global mask count per image
    res = aligned - target
    data_div = (2 * M).sum()
    total = (mask * res * res).sum() / data_div if float(data_div) != 0 else pred.new_zeros(())
    if alpha > 0:
        d = mask * res
        gx = (d[:, :, 1:] - d[:, :, :-1]).abs() * (mask[:, :, 1:] * mask[:, :, :-1])
        gy = (d[:, 1:, :] - d[:, :-1, :]).abs() * (mask[:, 1:, :] * mask[:, :-1, :])
        reg_div = M.sum()
        reg = (gx.sum() + gy.sum()) / reg_div if float(reg_div) != 0 else pred.new_zeros(())
        total = total + alpha * reg
    return total / weight


class SLAMLoss(nn.Module):
    def __init__(self, rgb_loss, eikonal_weight, trainer=None, train_dataset=None, assign_scale_shift_init=False,
                 smooth_weight=0.005, warp_loss_type="l1", depth_weight=0.1, normal_l1_weight=0.05,
                 normal_cos_weight=0.05, gt_depth_weight=0.0, flow_weight=0.0, warp_loss_weight=0, scan_id=-1,
                 model=None, rgb_loss_weight=1.0, assign_scale=20.0):
        super().__init__()
        if warp_loss_type not in ("l1", "ssim"):
            raise NotImplementedError("Strange patch loss type")
        self.model, self.trainer, self.train_dataset, self.scan_id = model, trainer, train_dataset, scan_id
        self.assign_scale_shift_init, self.assign_scale = assign_scale_shift_init, assign_scale
        self.eikonal_weight, self.smooth_weight, self.depth_weight = eikonal_weight, smooth_weight, depth_weight
        self.normal_l1_weight, self.normal_cos_weight = normal_l1_weight, normal_cos_weight
        self.gt_depth_weight, self.flow_weight = gt_depth_weight, flow_weight
        self.warp_loss_weight, self.warp_loss_type, self.rgb_loss_weight = warp_loss_weight, warp_loss_type, rgb_loss_weight
        self.rgb_loss = get_class(rgb_loss)(reduction="mean") if isinstance(rgb_loss, str) else rgb_loss
        self._ssim = {}

    # ---- individual terms (same names as the reference so callers/plots can reuse them) ----
    def get_rgb_loss(self, rgb_values, rgb_gt, mask=None):
        a, b = rgb_values.reshape(-1, 3), rgb_gt.reshape(-1, 3)
        if mask is not None:
            m = mask.reshape(-1)
            a, b = a[m], b[m]
        return self.rgb_loss(a, b)

    def get_gt_depth_loss(self, depth_values, depth_gt, mask=None):
        return _masked_mean_abs(depth_values.reshape(-1, 1), depth_gt.reshape(-1, 1), mask)

    def get_eikonal_loss(self, grad_theta):
        return ((grad_theta.norm(2, dim=1) - 1) ** 2).mean()

    def get_smooth_loss(self, model_outputs):
        unit = lambda g: g / (g.norm(2, dim=1).unsqueeze(-1) + 1e-5)
        return torch.norm(unit(model_outputs["grad_theta"]) - unit(model_outputs["grad_theta_nei"]), dim=-1).mean()

    def get_depth_loss(self, depth_pred, depth_gt, mask, keyframe_list=None):
        # self.depth_shard = (process group, this rank's share of the ray batch) in ray-sharded multi-GPU mapping, else None
        return scale_shift_invariant_depth_loss(depth_pred, depth_gt * 50 + 0.5, mask, alpha=0.5,
                                                shard=getattr(self, "depth_shard", None))

    def get_normal_loss(self, normal_pred, normal_gt):
        g = torch.nn.functional.normalize(normal_gt, p=2, dim=-1)
        p = torch.nn.functional.normalize(normal_pred, p=2, dim=-1)
        return (p - g).abs().sum(dim=-1).mean(), (1.0 - (p * g).sum(dim=-1)).mean()

    def _kernels(self, t):
        """masked-L1 terms as HIP launches (fused/warp.py::masked_l1) for device tensors, unless engine == "torch"."""
        return getattr(self, "engine", "auto") != "torch" and t.is_cuda and t.dtype == torch.float32

    def get_flow_loss(self, model_outputs, ground_truth, keyframe_list=None):
        if "flow" not in model_outputs:
            return 0.0
        m = ground_truth["flow_mask"]
        flow = model_outputs["flow"]
        if self._kernels(flow):
            from ..fused.warp import masked_l1
            return masked_l1(flow, ground_truth["flow"], m, 2)
        return (flow[m] - ground_truth["flow"].to(flow.device)[m]).abs().mean()

    def _warp_loss(self, warp_output):
        total = 0.0
        for patchsize, (gt_rgb, sampled, mask, _ray_mask) in warp_output.items():
            if patchsize == 1 or self.warp_loss_type == "l1":
                if self._kernels(sampled):
                    from ..fused.warp import masked_l1
                    total = total + masked_l1(sampled, gt_rgb, mask, 3)
                else:
                    total = total + (sampled[mask] - gt_rgb[mask]).abs().mean()
            else:   # "ssim": needs pytorch_msssim, exactly as the reference does
                try:
                    from pytorch_msssim import SSIM
                except ImportError as e:
                    raise NotImplementedError("warp_loss_type='ssim' needs the pytorch_msssim package") from e
                if patchsize not in self._ssim:
                    self._ssim[patchsize] = SSIM(data_range=1, win_size=patchsize, size_average=True, channel=3)
                a = torch.where(mask[..., None], sampled, torch.zeros_like(sampled))
                b = torch.where(mask[..., None], gt_rgb, torch.zeros_like(gt_rgb))
                a = a.reshape(-1, patchsize, patchsize, 3).permute(0, 3, 1, 2)
                b = b.reshape(-1, patchsize, patchsize, 3).permute(0, 3, 1, 2)
                total = total + 0.05 * (1 - self._ssim[patchsize](a, b))
        return total

    def _fused_ok(self, model_outputs):
        """The HIP loss kernels (fused/loss.py) cover the per-ray terms when the outputs live on the GPU, the colour term is the
        configured L1 mean and no ray-sharded depth reduction is requested; ``self.engine = "torch"`` forces the torch ops."""
        if getattr(self, "engine", "auto") == "torch" or getattr(self, "depth_shard", None) is not None:
            return False
        from ..fused import loss as fl
        rl = self.rgb_loss
        return (fl.available(model_outputs["rgb_values"]) and isinstance(rl, nn.L1Loss) and rl.reduction == "mean"
                and "normal_map" in model_outputs and model_outputs["depth_values"].dim() == 3)

    def _forward_fused(self, model_outputs, ground_truth, keyframe_list, frame_idx, stage):
        from ..fused.loss import fused_terms
        dev = model_outputs["rgb_values"].device
        depth_gt, depth_real_gt = ground_truth["depth"].to(dev), ground_truth["gt_depth"].to(dev)
        warp_loss = 0.0
        if "warp_output" in model_outputs and self.warp_loss_weight > 0 and stage == "fine" and frame_idx != 0:
            warp_loss = self._warp_loss(model_outputs["warp_output"])
        use_eik = self.eikonal_weight > 0 and "grad_theta" in model_outputs
        use_smooth = self.smooth_weight > 0.0
        depth_real = depth_real_gt
        if self.assign_scale_shift_init:      # loss.py:179-185
            if frame_idx == 0:
                depth_real = depth_gt * self.assign_scale
                self.gt_depth_weight = 10
            else:
                self.gt_depth_weight = 0
        whole = (self.depth_weight > 0 and self.train_dataset is not None
                 and "Replica" in getattr(self.train_dataset, "data_dir", "") and self.scan_id == 4)
        normals = self.normal_l1_weight > 0 or self.normal_cos_weight > 0
        w = (self.rgb_loss_weight, self.eikonal_weight if use_eik else 0.0, self.smooth_weight if use_smooth else 0.0,
             self.depth_weight, self.gt_depth_weight, self.normal_l1_weight if normals else 0.0,
             self.normal_cos_weight if normals else 0.0)
        total, t = fused_terms(model_outputs, ground_truth["rgb"], depth_gt, depth_real, depth_real_gt, ground_truth["mask"],
                               ground_truth["normal"], w, whole, use_eik, use_smooth)
        flow_loss = self.get_flow_loss(model_outputs, ground_truth, keyframe_list) if self.flow_weight > 0.0 else 0.0
        loss = total + self.flow_weight * flow_loss + self.warp_loss_weight * warp_loss
        on = lambda flag, v: v if flag else 0.0
        return {"loss": loss, "normal_l1": on(normals, t[5]), "depth_loss": on(self.depth_weight > 0, t[3]),
                "normal_cos": on(normals, t[6]), "gt_depth_loss": on(self.gt_depth_weight > 0, t[4]),
                "flow_loss": self.flow_weight * flow_loss, "rgb_loss": self.rgb_loss_weight * t[0],
                "warp_loss": self.warp_loss_weight * warp_loss, "smooth_loss": self.smooth_weight * on(use_smooth, t[2]),
                "eikonal_loss": self.eikonal_weight * on(use_eik, t[1])}

    def _tracking_objective(self, model_outputs, ground_truth):
        """The scalar the fused forward already formed (fused/track_graph.py: ``model_outputs["tracking_rgb_l1"]`` = (mean |rgb_values -
        gt|, gt)) when THIS loss is exactly that term: the reference's `tracking_loss` configuration (every weight but the colour term's
        zero, confs/*/runconf_*.conf `tracking_loss { ... }`; loss.py:131-233 then reduces to rgb_loss_weight * L1Loss(rgb_values, rgb))
        and the ground truth handed over is the very tensor the model formed it against.  Else None."""
        obj = model_outputs.get("tracking_rgb_l1")
        if obj is None or getattr(self, "engine", "auto") == "torch":
            return None
        rl = self.rgb_loss
        only_rgb = (isinstance(rl, nn.L1Loss) and rl.reduction == "mean" and not self.assign_scale_shift_init
                    and self.gt_depth_weight == 0 and self.depth_weight == 0 and self.normal_l1_weight == 0 and self.normal_cos_weight == 0
                    and self.smooth_weight == 0 and self.flow_weight == 0
                    and (self.eikonal_weight == 0 or "grad_theta" not in model_outputs)
                    and (self.warp_loss_weight == 0 or "warp_output" not in model_outputs))
        if not only_rgb or obj[1] is not ground_truth.get("rgb"):
            return None
        return obj[0]

    def forward(self, model_outputs, ground_truth, keyframe_list=None, frame_idx=0, stage="coarse"):
        obj = self._tracking_objective(model_outputs, ground_truth)
        if obj is not None:      # one tensor, no launch: the value and the gradient were formed by the model's own forward
            rgb_loss = obj if self.rgb_loss_weight == 1.0 else self.rgb_loss_weight * obj
            return {"loss": rgb_loss, "normal_l1": 0.0, "depth_loss": 0.0, "normal_cos": 0.0, "gt_depth_loss": 0.0, "flow_loss": 0.0,
                    "rgb_loss": rgb_loss, "warp_loss": 0.0, "smooth_loss": 0.0, "eikonal_loss": 0.0}
        if self._fused_ok(model_outputs):
            return self._forward_fused(model_outputs, ground_truth, keyframe_list, frame_idx, stage)
        rgb_pred, depth_pred = model_outputs["rgb_values"], model_outputs["depth_values"]
        dev = rgb_pred.device
        rgb_gt, depth_gt = ground_truth["rgb"].to(dev), ground_truth["depth"].to(dev)
        normal_gt, depth_real_gt = ground_truth["normal"].to(dev), ground_truth["gt_depth"].to(dev)
        normal_pred = model_outputs["normal_map"][None]
        bs = depth_pred.shape[0]

        rgb_loss = self.get_rgb_loss(rgb_pred, rgb_gt)
        warp_loss = 0.0
        if "warp_output" in model_outputs and self.warp_loss_weight > 0 and stage == "fine" and frame_idx != 0:
            warp_loss = self._warp_loss(model_outputs["warp_output"])
        eikonal_loss = 0.0
        if self.eikonal_weight > 0 and "grad_theta" in model_outputs:
            eikonal_loss = self.get_eikonal_loss(model_outputs["grad_theta"])

        # foreground = rays whose samples straddle the surface (sdf changes sign), loss.py:164-167
        sdf = model_outputs["sdf"]
        mask = ((sdf > 0.0).any(dim=-1) & (sdf < 0.0).any(dim=-1)).reshape(bs, -1, 1)
        mask = (ground_truth["mask"].to(dev) > 0.5) & mask

        depth_loss = 0.0
        if self.depth_weight > 0:
            whole = (self.train_dataset is not None and "Replica" in getattr(self.train_dataset, "data_dir", "")
                     and self.scan_id == 4)
            depth_loss = self.get_depth_loss(depth_pred, depth_gt, torch.ones_like(depth_pred) if whole else mask,
                                             keyframe_list)
        if self.assign_scale_shift_init:      # first frame: supervise with the scaled monocular depth (loss.py:179-185)
            if frame_idx == 0:
                depth_real_gt = depth_gt * self.assign_scale
                self.gt_depth_weight = 10
            else:
                self.gt_depth_weight = 0
        gt_depth_loss = 0.0
        if self.gt_depth_weight > 0:
            gt_depth_loss = self.get_gt_depth_loss(depth_pred, depth_real_gt, ground_truth["gt_depth"].to(dev) > 0)
        normal_l1 = normal_cos = 0.0
        if self.normal_l1_weight > 0 or self.normal_cos_weight > 0:
            normal_l1, normal_cos = self.get_normal_loss(normal_pred * mask, normal_gt * mask)
        smooth_loss = self.get_smooth_loss(model_outputs) if self.smooth_weight > 0.0 else 0.0
        flow_loss = self.get_flow_loss(model_outputs, ground_truth, keyframe_list) if self.flow_weight > 0.0 else 0.0

        loss = (self.flow_weight * flow_loss + self.depth_weight * depth_loss + self.rgb_loss_weight * rgb_loss
                + self.smooth_weight * smooth_loss + self.normal_l1_weight * normal_l1 + self.warp_loss_weight * warp_loss
                + self.eikonal_weight * eikonal_loss + self.normal_cos_weight * normal_cos
                + self.gt_depth_weight * gt_depth_loss)
        return {"loss": loss, "normal_l1": normal_l1, "depth_loss": depth_loss, "normal_cos": normal_cos,
                "gt_depth_loss": gt_depth_loss, "flow_loss": self.flow_weight * flow_loss,
                "rgb_loss": self.rgb_loss_weight * rgb_loss, "warp_loss": self.warp_loss_weight * warp_loss,
                "smooth_loss": self.smooth_weight * smooth_loss, "eikonal_loss": self.eikonal_weight * eikonal_loss}
