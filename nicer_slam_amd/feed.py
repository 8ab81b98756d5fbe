"""Per-iteration input feed kept on the device.

The reference's dataset re-uploads the full RGB / mask / depth / normal / GT-depth images of every frame of the batch
with ``.cuda()`` on EVERY iteration and draws the pixel subset on the CPU (code/datasets/scene_dataset.py:214-257,
change_sampling_idx :277-287): ~23 MB of PCIe traffic per frame per iteration to use 1024-8192 pixels.  Here every
frame is uploaded once when it arrives and stays resident (680x1200: 26 MB per frame, 2000 frames = 52 GB of the
288 GB; colour and metric depth in two contiguous stores that the patch-warp kernels index by slot); one iteration draws pixel indices with the device generator and gathers uv / GT with index_select.

Output dictionaries have the reference's keys and shapes (collate_fn :259-275), so SLAMNetwork.forward and SLAMLoss
consume them unchanged.  Image decoding / file layout stay with the caller (dataset tooling is out of scope).
"""
import torch


class FrameFeed:
    def __init__(self, img_res, device="cuda", scene_scale=1.0, capacity=8):
        self.H, self.W = int(img_res[0]), int(img_res[1])
        self.total_pixels = self.H * self.W
        self.device = torch.device(device)
        self.scene_scale = float(scene_scale)
        self.frames = {}
        self.sampling_idx = None
        # full frames live in two stores ([capacity, H*W, 3] colour, [capacity, H*W, 1] metric depth already divided by
        # scene_scale); a frame's tensors are views of its slot, so the re-projection kernels (fused/warp.py) can index the
        # store by slot instead of receiving a per-iteration stacked copy (8 x 26 MB at 680x1200).
        self.capacity = int(capacity)
        self._store_rgb = torch.empty(self.capacity, self.total_pixels, 3, device=self.device)
        self._store_depth = torch.empty(self.capacity, self.total_pixels, 1, device=self.device)
        self._slot = {}
        self._free = list(range(self.capacity - 1, -1, -1))
        self._index_cache = {}
        # pixel grid in the reference's order: uv[i] = (i % W, i // W)   (scene_dataset.py:106-111)
        i = torch.arange(self.total_pixels, device=self.device)
        self.uv = torch.stack([(i % self.W).float(), (i // self.W).float()], -1)

    # ------------------------------------------------------------------ frame store
    def add_frame(self, idx, rgb, depth, normal, gt_depth=None, mask=None, intrinsics=None, pose=None):
        """Upload one frame (once).  rgb [H*W,3], depth (monocular) [H*W,1], normal [H*W,3], gt_depth [H*W,1] or None
        (-> ones, scene_dataset.py:204-205), mask [H*W,1] or None (-> ones), intrinsics [4,4], pose (estimate) [4,4]."""
        n = self.total_pixels
        to = lambda t, c: torch.as_tensor(t, dtype=torch.float32).reshape(n, c).to(self.device, non_blocking=True)
        depth = to(depth, 1)
        slot = self._take_slot(int(idx))
        self._store_rgb[slot].copy_(to(rgb, 3))
        self._store_depth[slot].copy_((to(gt_depth, 1) if gt_depth is not None else torch.ones_like(depth)) / self.scene_scale)
        self.frames[int(idx)] = {
            "rgb": self._store_rgb[slot], "depth": depth, "normal": to(normal, 3),
            "gt_depth": self._store_depth[slot],                          # already / scene_scale
            "mask": to(mask, 1) if mask is not None else torch.ones_like(depth),
            "intrinsics": torch.as_tensor(intrinsics, dtype=torch.float32).reshape(4, 4).to(self.device),
            "pose": torch.as_tensor(pose, dtype=torch.float32).reshape(4, 4).to(self.device).clone(),
        }

    def set_pose(self, idx, pose):
        self.frames[int(idx)]["pose"].copy_(torch.as_tensor(pose, dtype=torch.float32).reshape(4, 4))

    def drop_frame(self, idx):
        if self.frames.pop(int(idx), None) is not None:
            self._free.append(self._slot.pop(int(idx)))
            self._index_cache.clear()

    def _take_slot(self, idx):
        if idx in self._slot:
            return self._slot[idx]
        if not self._free:                                                # grow x2; frames re-point at the new storage
            old_cap = self.capacity
            self.capacity *= 2
            for name in ("_store_rgb", "_store_depth"):
                old = getattr(self, name)
                new = torch.empty(self.capacity, *old.shape[1:], device=self.device)
                new[:old_cap].copy_(old)
                setattr(self, name, new)
            for i, sl in self._slot.items():
                self.frames[i]["rgb"], self.frames[i]["gt_depth"] = self._store_rgb[sl], self._store_depth[sl]
            self._free = list(range(self.capacity - 1, old_cap - 1, -1))
            self._index_cache.clear()
        self._slot[idx] = self._free.pop()
        return self._slot[idx]

    def _slots_of(self, frame_ids):
        """int32 device tensor of the frames' store slots; cached per keyframe list (the local window is constant over the
        iterations of a mapping round, volsdf_train.py:474-490), so a batch uploads nothing."""
        key = tuple(int(i) for i in frame_ids)
        hit = self._index_cache.get(key)
        if hit is None:
            hit = torch.tensor([self._slot[i] for i in key], dtype=torch.int32, device=self.device)
            if len(self._index_cache) > 64:
                self._index_cache.clear()
            self._index_cache[key] = hit
        return hit

    # ------------------------------------------------------------------ sampling
    def change_sampling_idx(self, sampling_size, generator=None, total_pixels=None):
        """-1: whole image (visualisation).  Else one index set shared by all frames of the next batch, drawn on the
        device (the reference draws on the CPU and uploads)."""
        if sampling_size == -1:
            self.sampling_idx = None
        else:
            self.sampling_idx = torch.randint(total_pixels or self.total_pixels, (int(sampling_size),),
                                              device=self.device, generator=generator)
        return self.sampling_idx

    def batch(self, frame_ids, full="stack"):
        """(indices [b] long, model_input, ground_truth) for the frames ``frame_ids`` at the current sampling_idx.
        ``full``: how ground_truth['full_rgb'] / ['full_depth'] (consumed by the patch-warp block) are handed over --
        "stack": the reference's [b, H*W, C] tensors (a copy of every frame of the batch per call, as collate_fn makes);
        "store": fused/warp.py::FrameStore views of the resident stores + the batch's slot indices (no copy)."""
        fr = [self.frames[int(i)] for i in frame_ids]
        sel = self.sampling_idx
        pick = (lambda t: t) if sel is None else (lambda t: t.index_select(0, sel))
        stack = lambda key, f=pick: torch.stack([f(x[key]) for x in fr])
        b = len(fr)
        model_input = {"uv": pick(self.uv).unsqueeze(0).expand(b, -1, -1).contiguous(),
                       "intrinsics": torch.stack([x["intrinsics"] for x in fr]),
                       "pose": torch.stack([x["pose"] for x in fr])}
        gt = {"rgb": stack("rgb"), "mask": stack("mask"), "depth": stack("depth"), "normal": stack("normal"),
              "gt_depth": stack("gt_depth")}
        if sel is not None:
            model_input["sampling_idx"] = sel.unsqueeze(0).expand(b, -1)
            if full == "store":
                from .fused.warp import FrameStore
                slots = self._slots_of(frame_ids)
                gt["full_rgb"], gt["full_depth"] = FrameStore(self._store_rgb, slots), FrameStore(self._store_depth, slots)
            else:
                gt["full_rgb"] = torch.stack([x["rgb"] for x in fr])
                gt["full_depth"] = torch.stack([x["gt_depth"] for x in fr])
        return torch.as_tensor([int(i) for i in frame_ids], dtype=torch.long), model_input, gt
