"""Streaming regressor for video (BASELINE.json config #5: decoded frames -> crops -> POCO, bs=128).

The reference moves every crop through the host: cv2.warpAffine per detection, stack, one H2D copy per crop
(pocolib/core/tester.py:178-212, dataset/inference.py:72-135).  Here one *frame* crosses PCIe once
(uint8, 6.2 MB at 1080p) into a ring of pinned/device buffers on a copy stream, all people in it are cropped on
the GPU straight into the resident [B,3,224,224] batch tensor, the forward is a hipGraph replay on fixed
buffers, and only the packed SMPL record (pose 216 | betas 10 | cam 3 | var 24 | confidence 1 = 254 floats per crop,
written by the engine inside the graph: poco_outputs_t.record) returns.  Per batch: one H2D copy of the staging record, ONE crop
launch for the crops of all frames, one graph replay, one D2H copy.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from ._lib import check, lib

REC = 254          # poco_outputs_t.record: [pose 216 | betas 10 | cam 3 | var 24 | post-processed confidence 1]


class CropStream:
    def __init__(self, model, frame_hw: Tuple[int, int], batch: Optional[int] = None, ring: int = 8,
                 bbox_scale: float = 1.0, want_vertices: bool = False):
        self.m = model.finalize()
        self.B = B = int(batch or model.max_batch)
        assert self.B <= model.max_batch
        self.H, self.W = frame_hw
        self.scale = float(bbox_scale)
        self.dev = model.device
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.ring = ring
        self.h_frames = [torch.empty(self.H, self.W, 3, dtype=torch.uint8).pin_memory() for _ in range(ring)]
        self.d_frames = [torch.empty(self.H, self.W, 3, dtype=torch.uint8, device=self.dev) for _ in range(ring)]
        # device table of the ring's frame pointers: ONE crop launch per batch cuts every crop from d_frames[frame_idx[n]]
        self.frame_ptrs = torch.tensor([t.data_ptr() for t in self.d_frames], dtype=torch.int64, device=self.dev)
        self.ev_up = [torch.cuda.Event() for _ in range(ring)]       # upload of slot k finished
        self.ev_free = [None] * ring                                 # event of the run whose crop launch read slot k last
        self._up_seq = [0] * ring                                    # order of the uploads on the (in-order) copy stream
        self._seq = 0
        f = torch.float32
        # One staging record per batch, fields contiguous (SoA): boxes [B,4] | bbox_info [B,3] | focal [B] | scale [B] | center [B,2] |
        # frame index [B] (int32 bits).  The batch tensors of the forward are VIEWS of its device copy: one H2D copy per batch and
        # no device-side reshuffling.  One pinned buffer per in-flight run (indexed by `host_buf` like rec_h): run(i+1) is enqueued
        # while run(i)'s H2D copy may still sit behind a ~30 ms forward.
        self._off = {"boxes": 0, "bbox_info": 4 * B, "focal_length": 7 * B, "scale": 8 * B, "center": 9 * B, "fidx": 11 * B}
        nmeta = 12 * B
        self.meta_h = [torch.zeros(nmeta, dtype=f).pin_memory() for _ in range(2)]
        for mh in self.meta_h:
            mh[8 * B:9 * B] = 1.0                                      # scale of the rows beyond a partial batch: finite arithmetic in ignored rows
        self.ev_meta = [torch.cuda.Event() for _ in range(2)]         # H2D copy of meta_h[i] finished
        self.meta_d = torch.zeros(nmeta, device=self.dev, dtype=f)
        self.meta_d[8 * B:9 * B] = 1.0
        o = self._off
        self.batch = {"img": torch.zeros(B, 3, 224, 224, device=self.dev, dtype=f),
                      "bbox_info": self.meta_d[o["bbox_info"]:o["bbox_info"] + 3 * B].view(B, 3),
                      "focal_length": self.meta_d[o["focal_length"]:o["focal_length"] + B],
                      "scale": self.meta_d[o["scale"]:o["scale"] + B],
                      "center": self.meta_d[o["center"]:o["center"] + 2 * B].view(B, 2),
                      "orig_shape": torch.tensor([[self.H, self.W]], device=self.dev, dtype=f).repeat(B, 1)}
        self.boxes_d = self.meta_d[0:4 * B].view(B, 4)
        self.fidx_d = self.meta_d[o["fidx"]:o["fidx"] + B].view(torch.int32)
        self.out = self.m._alloc_outputs(B, False)
        self._pending = [False] * ring                                # slot uploaded and not yet consumed by run()
        self.rec_h = [torch.empty(B, REC, dtype=f).pin_memory() for _ in range(2)]
        self.want_vertices = want_vertices
        self._slot = 0
        self._pool, self._pool_n = None, 0
        self._L = lib()
        self._L.poco_crop_normalize_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                      C.c_double, C.c_int, C.c_void_p, C.c_void_p]
        self.focal = float((self.W ** 2 + self.H ** 2) ** 0.5)

    # -- one frame: async upload into the ring ----------------------------------------------------------
    def upload(self, frame: np.ndarray) -> int:
        k = self._slot
        if self._pending[k]:
            raise RuntimeError(f"CropStream ring of {self.ring} frames is too small: slot {k} would be re-uploaded before "
                               "run() consumed it (enqueue fewer frames ahead or build the stream with a larger ring)")
        self._slot = (k + 1) % self.ring
        self._pending[k] = True
        if self.ev_free[k] is not None:
            self.ev_free[k].synchronize()                   # host: the slot's previous crops were consumed
        self.h_frames[k].numpy()[...] = frame
        with torch.cuda.stream(self.copy_stream):
            self.d_frames[k].copy_(self.h_frames[k], non_blocking=True)
            self.ev_up[k].record(self.copy_stream)
        self._seq += 1
        self._up_seq[k] = self._seq
        return k

    def upload_many(self, frames: Sequence[np.ndarray], threads: int = 4) -> List[int]:
        """upload() for the frames of one batch with the host staging done in parallel: the copy of a decoded frame into its
        pinned ring slot (6.2 MB at 1080p, ~0.7 ms on one core = what bounded the config-#5 pipeline at ~1000 frames/s) runs on a
        small thread pool (numpy releases the GIL for the copy); the H2D copies are then enqueued in frame order.  A decoder that
        can write into a given buffer should skip the staging copy altogether: `pinned_frame(slot)` is the slot's host array."""
        from concurrent.futures import ThreadPoolExecutor
        if len(frames) > self.ring:
            raise RuntimeError(f"CropStream ring of {self.ring} frames is too small for {len(frames)} frames at once")
        slots = [(self._slot + i) % self.ring for i in range(len(frames))]
        busy = [k for k in slots if self._pending[k]]
        if busy:            # checked before any state changes: an overflow must not leave earlier slots reserved for ever
            raise RuntimeError(f"CropStream ring of {self.ring} frames is too small for {len(frames)} more frames "
                               f"(slots {busy} not yet consumed by run())")
        for k in slots:
            self._pending[k] = True
        self._slot = (self._slot + len(frames)) % self.ring
        if self._pool is None or self._pool_n != threads:
            self._pool, self._pool_n = ThreadPoolExecutor(threads), threads

        def stage(k, frame):
            if self.ev_free[k] is not None:
                self.ev_free[k].synchronize()
            self.h_frames[k].numpy()[...] = frame

        try:
            list(self._pool.map(stage, slots, frames))
        except Exception:
            for k in slots:
                self._pending[k] = False
            raise
        with torch.cuda.stream(self.copy_stream):
            for k in slots:
                self.d_frames[k].copy_(self.h_frames[k], non_blocking=True)
                self.ev_up[k].record(self.copy_stream)
                self._seq += 1
                self._up_seq[k] = self._seq
        return slots

    def pinned_frame(self, slot: int) -> np.ndarray:
        """The pinned host array [H,W,3] uint8 of a ring slot (zero-copy staging for decoders that fill a caller's buffer)."""
        return self.h_frames[slot].numpy()

    def release(self, slot: int) -> None:
        """Give back a ring slot whose frame will not be handed to run() (a frame without detections, an aborted batch):
        upload() may reuse it.  run() releases the slots of its `groups` itself (also when it raises)."""
        self._pending[slot] = False

    # -- a full batch: (slot, [n,4] boxes) pairs with sum(n) <= B -----------------------------------------
    def run(self, groups: Sequence[Tuple[int, np.ndarray]], host_buf: int = 0,
            keep: Sequence[int] = ()) -> Tuple[torch.Tensor, int]:
        """Crops + forward + packed record D2H (async).  Returns (pinned host records, n); the caller must
        torch.cuda.current_stream().synchronize() (or wait on its own event) and then call check() before reading them.
        `host_buf` (0/1) selects the pinned staging buffers of this run; alternate it between consecutive runs that
        are in flight together.  `keep`: ring slots whose frame a later run() will crop from again (a frame with more
        people than fit into this batch); all other slots in `groups` become free for upload() once this run's
        crops are done.  A frame without detections must not stay uploaded: pass it with an empty box array (its slot is
        released, nothing is cropped) or call release(slot)."""
        try:
            return self._run(groups, host_buf, keep)
        finally:
            # whatever happened, the slots of this call are no longer waiting for a run(): a failed / asserting call must not
            # leave the ring guard set for ever; slots in `keep` stay reserved for the caller's next run()
            for slot, _ in groups:
                if slot not in keep:
                    self._pending[slot] = False

    def check(self) -> None:
        """Call after the event / stream synchronise that makes a run()'s records readable and BEFORE using them: raises if a bounded
        in-kernel wait timed out in one of the forwards since the last check (poco_status, include/poco_hip.h) - a graph replay never
        re-enters poco_forward, so nothing else would report it.  One read of host memory on the good path."""
        self.m.check_status()

    def _run(self, groups, host_buf, keep):
        st = torch.cuda.current_stream()
        groups = [(slot, boxes) for slot, boxes in groups if len(np.asarray(boxes).reshape(-1, 4))]     # empty box arrays: slot released only
        if not groups:
            return self.rec_h[host_buf], 0
        self.ev_meta[host_buf].synchronize()                # host: the previous H2D copy out of this staging buffer is done
        B, o = self.B, self._off
        mh = self.meta_h[host_buf].numpy()
        box_h = mh[0:4 * B].reshape(B, 4)
        info_h = mh[o["bbox_info"]:o["bbox_info"] + 3 * B].reshape(B, 3)
        cen_h = mh[o["center"]:o["center"] + 2 * B].reshape(B, 2)
        fidx_h = mh[o["fidx"]:o["fidx"] + B].view(np.int32)
        n = 0
        for slot, boxes in groups:
            b = np.asarray(boxes, np.float32).reshape(-1, 4)
            k = len(b)
            assert n + k <= B, "more crops than the stream's batch size"
            s = np.maximum(b[:, 2], b[:, 3]) / 200.0
            box_h[n:n + k] = b
            info_h[n:n + k, 0] = (b[:, 0] - self.W / 2.0) / self.focal * 2.8        # image_utils.py:174-187
            info_h[n:n + k, 1] = (b[:, 1] - self.H / 2.0) / self.focal * 2.8
            info_h[n:n + k, 2] = (s * 200.0 - 0.24 * self.focal) / (0.06 * self.focal)
            mh[o["focal_length"] + n:o["focal_length"] + n + k] = self.focal
            mh[o["scale"] + n:o["scale"] + n + k] = s
            cen_h[n:n + k] = b[:, 0:2]
            fidx_h[n:n + k] = slot
            n += k
        assert 0 < n <= B
        self.meta_d.copy_(self.meta_h[host_buf], non_blocking=True)     # the batch tensors are views of meta_d: nothing else to move
        self.ev_meta[host_buf].record(st)
        # the copy stream is in order: waiting for the most recent upload among this batch's slots covers the earlier ones
        last = max((slot for slot, _ in groups), key=lambda k: self._up_seq[k])
        st.wait_event(self.ev_up[last])
        check(self._L.poco_crop_normalize_multi(self.frame_ptrs.data_ptr(), self.ring, self.fidx_d.data_ptr(), self.H, self.W,
                                                self.boxes_d.data_ptr(), n, self.scale, 224, self.batch["img"].data_ptr(),
                                                C.c_void_p(st.cuda_stream)), "poco_crop_normalize_multi")
        ev = torch.cuda.Event()
        ev.record(st)
        for slot, _ in groups:
            self.ev_free[slot] = ev                           # upload() of these slots waits for this batch's crop launch
        out = self.m.graph_forward(self.batch, self.out)      # full-B replay; rows >= n are stale crops, ignored
        h = self.rec_h[host_buf]
        h.copy_(out["record"], non_blocking=True)             # the packed record is written by the engine inside the graph
        return h, n
