"""Device-resident frame feed on the GPU vs batches captured from the reference's own dataset class."""
import pytest
import torch

from test_feed_cpu import check_feed_against_reference

pytestmark = pytest.mark.gpu


def test_frame_feed_on_device_vs_reference_dataset_batches():
    check_feed_against_reference("cuda")


def test_frame_feed_drives_a_fused_tracking_forward():
    """feed.batch() output goes straight into SLAMNetwork.forward (fused engine) -- the per-iteration path of
    volsdf_train.py:411-417 with the frames already resident."""
    from nicer_slam_amd.feed import FrameFeed
    from nicer_slam_amd.model.network import SLAMNetwork
    from nicer_slam_amd.utils.conf import replica_model_conf
    torch.manual_seed(0)
    H, W = 68, 120
    model = SLAMNetwork(replica_model_conf(use_warp_loss=False), n_images=1,
                        colour_grid=dict(base_resolution=16, desired_resolution=256, log2_hashmap_size=14)).cuda().train()
    for p in model.parameters():
        p.requires_grad_(False)
    feed = FrameFeed((H, W), device="cuda")
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = 60.0
    K[0, 2], K[1, 2] = W / 2 - 0.5, H / 2 - 0.5
    pose = torch.eye(4)
    pose[:3, 3] = torch.tensor([0.1, 0.0, -0.2])
    feed.add_frame(0, rgb=torch.rand(H * W, 3), depth=torch.rand(H * W, 1), normal=torch.rand(H * W, 3), intrinsics=K, pose=pose)
    feed.change_sampling_idx(256)
    indices, inp, gt = feed.batch([0])
    out = model(inp, indices, gt, mode="tracking", frame_idx=0)
    assert model.last_engine == "fused"
    assert out["rgb_values"].shape == (1, 256, 3) and gt["rgb"].shape == (1, 256, 3)
    assert bool(torch.isfinite(out["rgb_values"]).all())


def test_feed_gather_is_one_launch_of_index_selects_and_survives_store_growth():
    """nsa_feed_gather (every field of every frame of the batch in one launch) against per-frame index_select on the frames' own
    tensors: bit-exact, also after the stores grew past their initial capacity and a frame was dropped and its slot reused; an index
    outside the image yields NaN rows (torch would assert)."""
    from nicer_slam_amd.feed import FrameFeed
    H, W = 17, 23
    g = torch.Generator().manual_seed(0)
    feed = FrameFeed((H, W), device="cuda", scene_scale=1.0, capacity=2)
    src = {}
    for idx in (5, 9, 2, 11, 7):                                       # capacity 2 -> 4 -> 8
        src[idx] = dict(rgb=torch.rand(H * W, 3, generator=g), depth=torch.rand(H * W, 1, generator=g),
                        normal=torch.rand(H * W, 3, generator=g), gt_depth=torch.rand(H * W, 1, generator=g),
                        mask=(torch.rand(H * W, 1, generator=g) > 0.3).float(), intrinsics=torch.eye(4), pose=torch.eye(4))
        feed.add_frame(idx, **src[idx])
    feed.drop_frame(9)
    src[4] = dict(src[5], rgb=torch.rand(H * W, 3, generator=g))
    feed.add_frame(4, **src[4])                                        # reuses the freed slot
    sel = feed.change_sampling_idx(301, generator=torch.Generator(device="cuda").manual_seed(3))
    ids = [7, 4, 5, 11, 2]
    indices, inp, gt = feed.batch(ids)
    assert indices.is_cuda and indices.tolist() == ids
    for i, fid in enumerate(ids):
        for k in ("rgb", "depth", "normal", "gt_depth", "mask"):
            assert torch.equal(gt[k][i].cpu(), src[fid][k][sel.cpu()]), (fid, k)
        assert torch.equal(inp["uv"][i], feed.uv[sel])
    feed.sampling_idx = torch.tensor([0, H * W, 5, -1], device="cuda")
    _, inp, gt = feed.batch([2])
    bad = torch.tensor([False, True, False, True])
    assert torch.equal(torch.isnan(gt["rgb"][0]).all(-1).cpu(), bad) and torch.equal(torch.isnan(inp["uv"][0]).all(-1).cpu(), bad)
    assert torch.equal(gt["normal"][0, 2].cpu(), src[2]["normal"][5])
