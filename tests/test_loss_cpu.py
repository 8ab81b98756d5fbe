"""SLAMLoss restatement (nicer_slam_amd/model/loss.py) vs goldens captured from the reference's SLAMLoss
(tests/golden/make_golden.py::loss_case): every returned term and the gradient w.r.t. every model output."""
import numpy as np
import pytest
import torch

from helpers import load, tt, assert_close


class _DS:
    data_dir = "../Datasets/processed/Replica"


def check_slam_loss(name, device="cpu", atol=1e-6, gtol=1e-7, engine="auto"):
    """shared with tests/test_loss_gpu.py (the same goldens with every tensor on the device)"""
    from nicer_slam_amd.model.loss import SLAMLoss
    fx = load(name)
    dv = lambda k: tt(fx[k]).to(device)
    leaf = lambda k: dv(k).clone().requires_grad_(True)
    out = {k: leaf("in_" + k) for k in ("rgb_values", "depth_values", "normal_map", "grad_theta", "grad_theta_nei", "flow")}
    out["sdf"] = dv("in_sdf")
    warp = {}
    for ps in (1, 5):
        warp[ps] = (dv(f"in_warp{ps}_gt"), leaf(f"in_warp{ps}_sampled"), dv(f"in_warp{ps}_mask").bool(),
                    dv(f"in_warp{ps}_raymask").bool())
    out["warp_output"] = warp
    gt = {k[3:]: dv(k) for k in fx if k.startswith("gt_")}
    gt["flow_mask"] = gt["flow_mask"].bool()
    if "meta_data_dir" in fx:        # a second conf family: its loss block comes from the shipped preset (utils/conf.py::run_conf)
        from nicer_slam_amd.utils.conf import run_conf
        rc = next(c for c in map(run_conf, ("replica", "7scenes", "azure")) if c["data_dir"] == str(fx["meta_data_dir"]))
        assert rc["loss"]["smooth_weight"] == float(fx["meta_smooth_weight"])

        class ds:
            data_dir = rc["data_dir"]
        crit = SLAMLoss(train_dataset=ds(), scan_id=1, **rc["loss"])
    else:
        crit = SLAMLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, train_dataset=_DS(), scan_id=1,
                        assign_scale_shift_init=True, smooth_weight=0.005, warp_loss_type="l1", depth_weight=0.1,
                        normal_l1_weight=0.05, normal_cos_weight=0.05, flow_weight=0.001, warp_loss_weight=0.5)
    crit.engine = engine
    assert crit._fused_ok(out) == (device == "cuda" and engine == "auto")      # GPU: the HIP loss kernels, not the torch ops
    res = crit(out, gt, keyframe_list=None, frame_idx=int(fx["meta_frame_idx"]), stage=str(fx["meta_stage"]))
    assert set(res) == {k[4:] for k in fx if k.startswith("out_")}
    for k, v in res.items():
        assert_close(torch.as_tensor(float(v)), fx["out_" + k], atol, 1e-5, k)
    res["loss"].backward()
    for k in ("rgb_values", "depth_values", "normal_map", "grad_theta", "grad_theta_nei", "flow"):
        if "grad_" + k in fx:
            assert_close(out[k].grad, fx["grad_" + k], gtol, 1e-4, "d/d " + k)
        else:
            assert out[k].grad is None or float(out[k].grad.abs().max()) == 0
    for ps in (1, 5):
        key = f"grad_warp{ps}_sampled"
        if key in fx:
            assert_close(warp[ps][1].grad, fx[key], gtol, 1e-4, key)


@pytest.mark.parametrize("name", ["loss_mapping_first_frame", "loss_mapping_fine", "loss_mapping_7scenes", "loss_mapping_azure_first_frame"])
def test_slam_loss_vs_reference_golden(name):
    check_slam_loss(name)


def test_depth_loss_degenerate_mask_is_zero():
    from nicer_slam_amd.model.loss import scale_shift_invariant_depth_loss
    p = torch.rand(2, 5, 1, requires_grad=True)
    out = scale_shift_invariant_depth_loss(p, torch.rand(2, 5, 1), torch.zeros(2, 5, 1))
    assert float(out) == 0.0
