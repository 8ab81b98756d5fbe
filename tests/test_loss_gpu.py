"""Row f1 on the device: SLAMLoss with every tensor on the GPU -- the fused HIP loss kernels (csrc/loss_terms.hip) and the torch
restatement -- against the goldens captured from the reference's SLAMLoss (code/model/loss.py:113-233): every returned term and
the gradient w.r.t. every model output."""
import pytest
import torch

from test_loss_cpu import check_slam_loss

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("engine", ["auto", "torch"])
@pytest.mark.parametrize("name", ["loss_mapping_first_frame", "loss_mapping_fine", "loss_mapping_7scenes", "loss_mapping_azure_first_frame"])
def test_slam_loss_on_device_vs_reference_golden(name, engine):
    """engine "auto" = the fused HIP loss kernels (nsa_slam_loss), "torch" = the torch restatement on the device"""
    check_slam_loss(name, device="cuda", atol=2e-6, gtol=2e-7, engine=engine)


class _Replica4:
    data_dir = "../Datasets/processed/Replica"


def _random_case(bs, n, S, E, seed, fg_frac=0.7):
    g = torch.Generator().manual_seed(seed)
    R = bs * n
    rn = lambda *s: torch.randn(*s, generator=g)
    sdf = rn(R, S).abs() + 0.01
    cross = torch.rand(R, generator=g) < fg_frac                    # rays whose sdf changes sign
    sdf[cross, S // 2:] *= -1
    leaf = lambda t: t.cuda().requires_grad_(True)
    out = {"rgb_values": leaf(torch.rand(bs, n, 3, generator=g)), "depth_values": leaf(torch.rand(bs, n, 1, generator=g) * 3 + 0.5),
           "normal_map": leaf(rn(bs, n, 3)), "grad_theta": leaf(rn(E, 3)), "grad_theta_nei": leaf(rn(E, 3)), "sdf": sdf.cuda()}
    gt = {"rgb": torch.rand(bs, n, 3, generator=g).cuda(), "depth": (torch.rand(bs, n, 1, generator=g) * 0.05).cuda(),
          "normal": rn(bs, n, 3).cuda(), "gt_depth": (torch.rand(bs, n, 1, generator=g) * 3 * (torch.rand(bs, n, 1, generator=g) > 0.2)).cuda(),
          "mask": (torch.rand(bs, n, 1, generator=g) > 0.1).float().cuda()}
    return out, gt


@pytest.mark.parametrize("variant", ["mapping_8192", "first_frame", "whole_image", "no_foreground", "no_smooth_no_eikonal"])
def test_fused_loss_kernels_vs_torch_restatement(variant):
    """The HIP loss kernels against the (golden-pinned) torch restatement on the device, at the mapping batch shape and on the
    branches the goldens do not take: the Replica-scan-4 whole-image depth mask, an empty foreground (depth term exactly 0),
    disabled terms."""
    from nicer_slam_amd.model.loss import SLAMLoss
    from helpers import assert_close
    bs, n, S, E = (8, 1024, 98, 22 * 8192) if variant == "mapping_8192" else (3, 200, 40, 4400)
    out, gt = _random_case(bs, n, S, E, seed=len(variant), fg_frac=0.0 if variant == "no_foreground" else 0.7)
    kw = dict(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, smooth_weight=0.005, depth_weight=0.1, normal_l1_weight=0.05,
              normal_cos_weight=0.05, assign_scale_shift_init=variant == "first_frame")
    if variant == "whole_image":
        kw.update(train_dataset=_Replica4(), scan_id=4)
    if variant == "no_smooth_no_eikonal":
        kw.update(eikonal_weight=0, smooth_weight=0.0)
    frame = 0 if variant == "first_frame" else 7
    results = {}
    for engine in ("auto", "torch"):
        crit = SLAMLoss(**kw)
        crit.engine = engine
        for v in out.values():
            v.grad = None
        res = crit(out, gt, keyframe_list=None, frame_idx=frame, stage="fine")
        res["loss"].backward()
        results[engine] = ({k: float(v) for k, v in res.items()},
                           {k: (v.grad.clone() if v.grad is not None else None) for k, v in out.items() if v.requires_grad})
    (ta, ga), (tt_, gt_) = results["auto"], results["torch"]
    assert set(ta) == set(tt_)
    for k in ta:
        assert_close(torch.tensor(ta[k]), torch.tensor(tt_[k]), 1e-6, 2e-5, k)
    if variant == "no_foreground":
        assert ta["depth_loss"] == 0.0
    for k in ga:
        if gt_[k] is None:
            assert ga[k] is None or float(ga[k].abs().max()) == 0, k
        else:
            assert_close(ga[k], gt_[k], 1e-9, 2e-4, "d/d " + k)
