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
