"""Device-resident frame feed (nicer_slam_amd/feed.py): shapes/keys of the reference's collate_fn, pixel order, gathers."""
import torch


def test_frame_feed_batches_match_reference_layout():
    from nicer_slam_amd.feed import FrameFeed
    H, W = 6, 10
    feed = FrameFeed((H, W), device="cpu", scene_scale=2.0)
    g = torch.Generator().manual_seed(0)
    frames = {}
    for idx in (3, 8):
        frames[idx] = dict(rgb=torch.rand(H * W, 3, generator=g), depth=torch.rand(H * W, 1, generator=g),
                           normal=torch.rand(H * W, 3, generator=g), gt_depth=torch.rand(H * W, 1, generator=g) * 4,
                           intrinsics=torch.eye(4) * (idx + 1), pose=torch.eye(4) + idx)
        feed.add_frame(idx, **frames[idx])
    assert feed.uv.shape == (H * W, 2)
    assert feed.uv[W + 3].tolist() == [3.0, 1.0]                         # (x, y) = (i % W, i // W)
    sel = feed.change_sampling_idx(7, generator=torch.Generator().manual_seed(1))
    assert sel.shape == (7,) and int(sel.max()) < H * W
    indices, inp, gt = feed.batch([8, 3])
    assert indices.tolist() == [8, 3] and indices.dtype == torch.long
    assert inp["uv"].shape == (2, 7, 2) and inp["intrinsics"].shape == (2, 4, 4) and inp["pose"].shape == (2, 4, 4)
    assert torch.equal(inp["uv"][0], feed.uv[sel]) and torch.equal(inp["uv"][1], feed.uv[sel])
    assert torch.equal(inp["sampling_idx"][0], sel)
    assert set(gt) == {"rgb", "mask", "depth", "normal", "gt_depth", "full_rgb", "full_depth"}
    assert torch.equal(gt["rgb"][0], frames[8]["rgb"][sel]) and torch.equal(gt["normal"][1], frames[3]["normal"][sel])
    assert torch.allclose(gt["gt_depth"][1], frames[3]["gt_depth"][sel] / 2.0)
    assert gt["full_rgb"].shape == (2, H * W, 3) and torch.allclose(gt["full_depth"][0], frames[8]["gt_depth"] / 2.0)
    assert torch.equal(gt["mask"], torch.ones(2, 7, 1))
    assert torch.equal(inp["pose"][0], torch.eye(4) + 8)
    feed.set_pose(8, torch.eye(4) * 5)
    assert torch.equal(feed.batch([8])[1]["pose"][0], torch.eye(4) * 5)
    feed.change_sampling_idx(-1)                                         # visualisation: whole image, no full_* keys
    _, inp, gt = feed.batch([3])
    assert inp["uv"].shape == (1, H * W, 2) and "sampling_idx" not in inp and "full_rgb" not in gt
    assert torch.equal(gt["rgb"][0], frames[3]["rgb"])
