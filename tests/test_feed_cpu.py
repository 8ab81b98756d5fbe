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


def check_feed_against_reference(device):
    """FrameFeed vs batches produced by the reference's own SLAMDataset.__getitem__ / collate_fn
    (tests/golden/make_golden.py::feed_case; scene_dataset.py:214-287): same keys, shapes, dtypes and values for a mapping
    batch (2 frames), a tracking batch and a whole-image visualisation batch, given the same pixel draw."""
    from helpers import load, tt
    from nicer_slam_amd.feed import FrameFeed
    fx = load("feed_batches")
    H, W, Hedge, Wedge = [int(v) for v in fx["meta_res"]]
    feed = FrameFeed((H, W), device=device, scene_scale=float(fx["meta_scene_scale"]))
    for idx in (3, 8):
        f = {k: tt(fx[f"frame{idx}_{k}"]) for k in ("rgb", "depth", "normal", "gt_depth", "mask", "intrinsics", "pose")}
        feed.add_frame(idx, **f)
    for tag in ("map", "trk", "vis"):
        frames = fx[f"{tag}_frames"].tolist()
        if f"{tag}_sampling_idx" in fx:
            feed.sampling_idx = tt(fx[f"{tag}_sampling_idx"]).to(device)
        else:
            feed.change_sampling_idx(-1)
        indices, inp, gt = feed.batch(frames)
        assert indices.tolist() == fx[f"{tag}_indices"].tolist() and indices.dtype == torch.long
        want_in = {k[len(tag) + 4:]: v for k, v in fx.items() if k.startswith(f"{tag}_in_")}
        want_gt = {k[len(tag) + 4:]: v for k, v in fx.items() if k.startswith(f"{tag}_gt_")}
        assert set(inp) == set(want_in) and set(gt) == set(want_gt), (tag, set(inp) ^ set(want_in), set(gt) ^ set(want_gt))
        for got, want in ((inp, want_in), (gt, want_gt)):
            for k, v in want.items():
                assert got[k].device.type == torch.device(device).type, k
                assert tuple(got[k].shape) == v.shape and got[k].dtype == tt(v).dtype, (tag, k, got[k].shape, v.shape)
                if k in ("gt_depth", "full_depth"):                    # x / scene_scale: torch's GPU kernel multiplies by 1/s
                    assert torch.allclose(got[k].cpu(), tt(v), rtol=2e-7, atol=0), (tag, k)
                else:
                    assert torch.equal(got[k].cpu(), tt(v)), (tag, k)  # pure gathers: bit-exact
    # the device-side draw: tracking samples the first tracking_total_pixels indices (scene_dataset.py:282-286)
    n_trk = (H - 2 * Hedge) * (W - 2 * Wedge)
    sel = feed.change_sampling_idx(4096, total_pixels=n_trk)
    assert sel.device.type == torch.device(device).type and int(sel.min()) >= 0 and int(sel.max()) == n_trk - 1
    assert len(set(sel.tolist())) == n_trk


def test_frame_feed_vs_reference_dataset_batches():
    check_feed_against_reference("cpu")


def test_frame_feed_store_growth_drop_and_slot_reuse():
    """Frames stay addressable while the stores grow past their capacity, a dropped frame's slot is reused, and a batch returns the
    right rows of the right frames (the CPU path of FrameFeed._gather: one flat index_select per field)."""
    from nicer_slam_amd.feed import FrameFeed
    H, W = 17, 23
    g = torch.Generator().manual_seed(0)
    feed = FrameFeed((H, W), device="cpu", capacity=2)
    src = {}
    for idx in (5, 9, 2, 11, 7):
        src[idx] = dict(rgb=torch.rand(H * W, 3, generator=g), depth=torch.rand(H * W, 1, generator=g),
                        normal=torch.rand(H * W, 3, generator=g), gt_depth=torch.rand(H * W, 1, generator=g),
                        mask=(torch.rand(H * W, 1, generator=g) > 0.3).float(), intrinsics=torch.eye(4), pose=torch.eye(4))
        feed.add_frame(idx, **src[idx])
    assert feed.capacity == 8
    feed.drop_frame(9)
    src[4] = dict(src[5], rgb=torch.rand(H * W, 3, generator=g))
    feed.add_frame(4, **src[4])
    assert len(set(feed._slot.values())) == 5
    sel = feed.change_sampling_idx(301, generator=torch.Generator().manual_seed(3))
    ids = [7, 4, 5, 11, 2]
    indices, inp, gt = feed.batch(ids)
    assert indices.tolist() == ids and indices.dtype == torch.long
    assert feed.batch(ids)[0] is indices                                 # cached: no per-iteration upload
    for i, fid in enumerate(ids):
        for k in ("rgb", "depth", "normal", "gt_depth", "mask"):
            assert torch.equal(gt[k][i], src[fid][k][sel]), (fid, k)
            assert torch.equal(feed.frames[fid][k], src[fid][k])
