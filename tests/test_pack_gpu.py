"""nsa_pack_blocks (csrc/map_tail.hip::k_pack_blocks: the gather plan + exact 3-way bf16 split of fused/pack.py in ONE launch) against
the torch restatement of the same plan (gather, .to(bfloat16) x 3, stack, cat, gather) that tests/test_pack_cpu.py holds to a numpy
emulation of the MFMA dataflow: every packed word of every layout must be identical.  Needs an MI355X."""
import pytest
import torch

from helpers import load
from test_model_cpu import build_model

pytestmark = pytest.mark.gpu


def _torch_pack(flat, blocks):
    from nicer_slam_amd.fused import pack
    ia, iv, perm = pack._plan(blocks, flat.device)
    words = torch.stack(pack.split_pieces(flat[ia]), 1).contiguous().view(torch.float32).reshape(-1)     # (the library's operand form)
    return torch.cat([words, flat[iv]])[perm]


@pytest.mark.parametrize("layout", ["sdf32_coarse", "sdf32_fine", "sdf16_coarse", "sdf16_fine", "colour"])
def test_pack_kernel_equals_the_torch_plan_bit_for_bit(layout):
    from nicer_slam_amd.fused import pack
    model = build_model(load("full_tracking_rw")).cuda()
    g = torch.Generator(device="cuda").manual_seed(4)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "embeddings" not in name and p.dtype == torch.float32:
                p.mul_(1.0 + 0.3 * torch.randn(p.shape, device="cuda", generator=g))
        # awkward values: zero, negative zero, tiny, huge
        model.implicit_network.coarse.lin0.bias[:4] = torch.tensor([0.0, -0.0, 1e-30, 3.0e20], device="cuda")
        imp = model.implicit_network
        if layout == "colour":
            net, (blocks, n) = model.rendering_network, pack.colour_net_index()
            flat = pack.flat_params(net)
            fn = pack.pack_colour_net
        else:
            net = imp.coarse if layout.endswith("coarse") else imp.fine
            NH, enc = net.num_layers - 2, net.encoding
            if layout.startswith("sdf16"):
                blocks, n = pack.sdf_net_index4(NH, enc.level_dim)
                fn = pack.pack_sdf_net4
            else:
                blocks, n = pack.sdf_net_index(NH, enc.num_levels, enc.level_dim)
                fn = pack.pack_sdf_net
            flat = pack.flat_params(net)
        want = _torch_pack(flat, blocks)
        got = fn(net)                                      # no_grad + device tensor: the kernel path
    assert got.shape == want.shape and got.dtype == torch.float32
    assert torch.equal(got.view(torch.int32), want.view(torch.int32)), int((got.view(torch.int32) != want.view(torch.int32)).sum())
    # and under autograd the differentiable torch path is still what runs
    got_g = fn(net)
    assert got_g.requires_grad and torch.equal(got_g.detach().view(torch.int32), want.view(torch.int32))
