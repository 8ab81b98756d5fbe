"""The library's fp32 GEMMs on the MI355X against float64 (csrc/mlp_common.hpp::NSA_FORM; tests/test_operand_form_cpu.py holds the
emulated arithmetic).  nsa_sdf_points -- both SDF networks, four GEMM layers, on 60 000 points -- is compared with the same networks
evaluated in float64 from the same fp32 inputs (positional encoding and the oracle's grid features promoted to double), next to the
yardstick: the oracle's plain fp32 torch evaluation (nn.Linear on the CPU, the reference's arithmetic).  The kernel must not be
further from float64 than that yardstick -- whichever operand form the library was built in."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mlp64(params, prefix, n_linear, h):
    for l in range(n_linear):
        g, v, b = (params[f"{prefix}.lin{l}.{k}"].double() for k in ("weight_g", "weight_v", "bias"))
        h = h @ (v * (g / v.norm(2, dim=1, keepdim=True))).T + b
        if l < n_linear - 1:
            h = torch.nn.functional.softplus(h, beta=100)
    return h


def test_sdf_values_are_no_further_from_float64_than_an_fp32_evaluation(capsys):
    from helpers import params_of, oracle_config
    from test_inference_gpu import _golden_model
    from oracle import render_ref as R
    from nicer_slam_amd import inference
    from nicer_slam_amd.fused import pack
    fx, model = _golden_model("full_vis_eval")
    cfg, params = oracle_config(fx), params_of(fx)
    g = torch.Generator().manual_seed(11)
    pts = (torch.rand(60000, 3, generator=g) * 2 - 1) * 0.999             # inside the cube: every level contributes
    with torch.no_grad():
        fp32 = R.sdf_vals(params, cfg, pts.clone(), "fine").reshape(-1).double()
        ref = torch.zeros(pts.shape[0], dtype=torch.float64)
        for prefix, spec in (("implicit_network.coarse", cfg.coarse), ("implicit_network.fine", cfg.fine)):
            feat = R.grid_features(pts / spec.divide_factor, params[prefix + ".encoding.embeddings"], spec.grid)
            h = torch.cat((R.positional_encoding(pts.double(), spec.multires), feat.double()), dim=-1)
            ref += _mlp64(params, prefix, spec.n_linear, h)[:, 0]
    got = torch.as_tensor(inference.sdf_values(model, pts.cuda(), "fine", chunk=20000)).detach().cpu().double().reshape(-1)
    e_k, e_32 = (got - ref).abs(), (fp32 - ref).abs()
    rms = lambda e: float((e ** 2).mean().sqrt())
    with capsys.disabled():
        print(f"\n  operand form {pack.operand_form()}: |sdf - float64|  kernel rms {rms(e_k):.2e} max {float(e_k.max()):.2e}   "
              f"fp32 torch rms {rms(e_32):.2e} max {float(e_32.max()):.2e}   (|sdf| rms {float(ref.pow(2).mean().sqrt()):.3f})")
    # (the kernel's own fp32 positional encoding and grid blend differ from torch's in the last bit: that is part of e_k and not of
    #  e_32, which starts from the very inputs the float64 evaluation was given -- hence the factors)
    assert rms(e_k) <= 1.5 * rms(e_32) and float(e_k.max()) <= 3.0 * float(e_32.max()), (rms(e_k), rms(e_32))
