"""Per-iteration input feed kept on the device.

The reference's dataset re-uploads the full RGB / mask / depth / normal / GT-depth images of every frame of the batch
with ``.cuda()`` on EVERY iteration and draws the pixel subset on the CPU (code/datasets/scene_dataset.py:214-257,
change_sampling_idx :277-287): ~23 MB of PCIe traffic per frame per iteration to use 1024-8192 pixels.  Here every
frame is uploaded once when it arrives and stays resident (680x1200: 26 MB per frame, 2000 frames = 52 GB of the
288 GB; every image kind in one contiguous store [capacity, H*W, C] that kernels index by slot); one iteration draws pixel
indices with the device generator and gathers uv and all ground-truth fields of all frames of the batch in ONE launch
(nsa_feed_gather, csrc/feed_gather.hip; on CPU tensors: one index_select per field).

Output dictionaries have the reference's keys and shapes (collate_fn :259-275), so SLAMNetwork.forward and SLAMLoss
consume them unchanged.  Image decoding / file layout stay with the caller (dataset tooling is out of scope).
"""
import torch


class FrameFeed:
    FIELDS = (("rgb", 3), ("depth", 1), ("normal", 3), ("gt_depth", 1), ("mask", 1))     # ground-truth keys of collate_fn, channels

    @property
    def _store_rgb(self):
        return self._stores["rgb"]

    @property
    def _store_depth(self):
        return self._stores["gt_depth"]

    def __init__(self, img_res, device="cuda", scene_scale=1.0, capacity=8):
        self.H, self.W = int(img_res[0]), int(img_res[1])
        self.total_pixels = self.H * self.W
        self.device = torch.device(device)
        self.scene_scale = float(scene_scale)
        self.frames = {}
        self.sampling_idx = None
        # full frames live in one store per image kind ([capacity, H*W, C]; metric depth already divided by scene_scale); a
        # frame's tensors are views of its slot, so the batch gather (below) and the re-projection kernels (fused/warp.py) index
        # the stores by slot instead of receiving a per-iteration stacked copy (8 x 26 MB at 680x1200).
        self.capacity = int(capacity)
        self._stores = {k: torch.empty(self.capacity, self.total_pixels, c, device=self.device) for k, c in self.FIELDS}
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
        st = self._stores
        st["rgb"][slot].copy_(to(rgb, 3))
        st["depth"][slot].copy_(depth)
        st["normal"][slot].copy_(to(normal, 3))
        st["gt_depth"][slot].copy_((to(gt_depth, 1) if gt_depth is not None else torch.ones_like(depth)) / self.scene_scale)
        if mask is not None:
            st["mask"][slot].copy_(to(mask, 1))
        else:
            st["mask"][slot].fill_(1.0)
        self.frames[int(idx)] = {
            **{k: st[k][slot] for k, _ in self.FIELDS},                   # views of the slot; gt_depth already / scene_scale
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
            for name, old in list(self._stores.items()):
                new = torch.empty(self.capacity, *old.shape[1:], device=self.device)
                new[:old_cap].copy_(old)
                self._stores[name] = new
            for i, sl in self._slot.items():
                for name, _ in self.FIELDS:
                    self.frames[i][name] = self._stores[name][sl]
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

    def _indices_of(self, frame_ids):
        """The batch's frame ids as a long tensor ON THE FEED'S DEVICE, cached per keyframe list like the slots: a per-iteration
        `.cuda()` of a fresh host tensor (what the reference's loop does with collate_fn's indices) is a pageable upload, i.e. a full
        device synchronisation at the head of every iteration -- it kept the host from running ahead and left the GPU idle for 1 ms of
        every 14.5-ms mapping iteration (profiles/r04_mapping_host.txt)."""
        key = ("ids",) + tuple(int(i) for i in frame_ids)
        hit = self._index_cache.get(key)
        if hit is None:
            hit = torch.tensor(key[1:], dtype=torch.long, device=self.device)
            if len(self._index_cache) > 64:
                self._index_cache.clear()
            self._index_cache[key] = hit
        return hit

    def _gather(self, frame_ids, sel):
        """uv [b,n,2] and the ground-truth fields [b,n,C] of the batch's frames at the pixel indices ``sel`` [n]."""
        slots = self._slots_of(frame_ids)
        b, n = len(slots), int(sel.shape[0])
        if self.device.type == "cuda":
            import ctypes
            from ._native import lib, check, FeedField
            # the kernel reads `sel` as int64 DEVICE memory: a CPU randperm (what the reference assigns to sampling_idx,
            # scene_dataset.py:296-300) or an int32 index is converted here instead of faulting on the device
            if sel.device != self.device or sel.dtype != torch.int64:
                sel = sel.to(device=self.device, dtype=torch.int64)
            sel = sel.contiguous()
            if slots.device != self.device:
                slots = slots.to(self.device)
            uv = torch.empty(b, n, 2, device=self.device)
            gt = {k: torch.empty(b, n, c, device=self.device) for k, c in self.FIELDS}
            fields = (FeedField * len(self.FIELDS))(*[FeedField(self._stores[k].data_ptr(), gt[k].data_ptr(), c) for k, c in self.FIELDS])
            with torch.cuda.device(self.device):       # (a feed on a non-current GPU launches on ITS device's current stream)
                check(lib.nsa_feed_gather(fields, len(self.FIELDS), slots.data_ptr(), b, sel.data_ptr(), n, self.total_pixels, self.W,
                                          uv.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
            return uv, gt
        flat = (slots.long().unsqueeze(1) * self.total_pixels + sel.unsqueeze(0)).reshape(-1)
        gt = {k: self._stores[k].view(-1, c).index_select(0, flat).view(b, n, c) for k, c in self.FIELDS}
        return self.uv.index_select(0, sel).unsqueeze(0).expand(b, -1, -1).contiguous(), gt

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
        b = len(fr)
        if sel is None:                                                   # whole images (visualisation)
            uv = self.uv.unsqueeze(0).expand(b, -1, -1).contiguous()
            gt = {k: torch.stack([x[k] for x in fr]) for k, _ in self.FIELDS}
        else:
            uv, gt = self._gather(frame_ids, sel)
        model_input = {"uv": uv, "intrinsics": torch.stack([x["intrinsics"] for x in fr]),
                       "pose": torch.stack([x["pose"] for x in fr])}
        if sel is not None:
            model_input["sampling_idx"] = sel.unsqueeze(0).expand(b, -1)
            if full == "store":
                from .fused.warp import FrameStore
                slots = self._slots_of(frame_ids)
                gt["full_rgb"], gt["full_depth"] = FrameStore(self._store_rgb, slots), FrameStore(self._store_depth, slots)
            else:
                gt["full_rgb"] = torch.stack([x["rgb"] for x in fr])
                gt["full_depth"] = torch.stack([x["gt_depth"] for x in fr])
        return self._indices_of(frame_ids), model_input, gt
