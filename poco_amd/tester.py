"""POCOTester for the MI355X engine: batch assembly + post-processing of
pocolib/core/tester.py:153-245 (folder mode) and :362-479 (per-track video mode), without the
third-party detector / tracker / renderer (out of scope: SURVEY.md 2 rows 2,27).  Detections are an
input: {image name: [[cx, cy, w, h], ...]} (what multi_person_tracker hands to the reference).

Per image the crops are produced on the GPU (poco_crop_normalize) from the decoded frame, so the
reference's per-detection cv2.warpAffine + per-crop H2D copy (tester.py:182-203) disappears."""
from __future__ import annotations

import ctypes as C
import json
import os
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import postproc
from ._lib import check, lib
from .config import model_kwargs, update_hparams
from .model import POCO

IMG_EXT = (".png", ".jpg", ".jpeg")


def calculate_focal_length(img_h, img_w):
    return float((img_w ** 2 + img_h ** 2) ** 0.5)          # image_utils.py:171-172


def calculate_bbox_info(center, scale, orig_shape):
    img_h, img_w = orig_shape
    f = calculate_focal_length(img_h, img_w)
    info = np.array([center[0] - img_w / 2.0, center[1] - img_h / 2.0, scale * 200.0])
    info[:2] = info[:2] / f * 2.8
    info[2] = (info[2] - 0.24 * f) / (0.06 * f)
    return info.astype(np.float32)                            # image_utils.py:174-187


def crop_normalize(frame_u8: torch.Tensor, boxes: torch.Tensor, bbox_scale: float = 1.0, res: int = 224) -> torch.Tensor:
    """frame uint8 [H,W,3] RGB cuda, boxes [N,4] (cx,cy,w,h) cuda float32 or float64 -> [N,3,res,res] fp32: the normalised
    crop of get_single_image_crop_demo (vibe_image_utils.py:233-266), byte-exact with cv2's fixed-point warpAffine."""
    assert frame_u8.is_cuda and frame_u8.dtype == torch.uint8 and frame_u8.is_contiguous()
    assert boxes.is_cuda and boxes.dtype in (torch.float32, torch.float64) and boxes.shape[1:] == (4,)
    H, W, _ = frame_u8.shape
    N = boxes.shape[0]
    out = torch.empty(N, 3, res, res, device=frame_u8.device, dtype=torch.float32)
    L = lib()
    fn = L.poco_crop_normalize if boxes.dtype == torch.float32 else L.poco_crop_normalize_f64
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    check(fn(frame_u8.data_ptr(), H, W, boxes.contiguous().data_ptr(), N, float(bbox_scale), res, out.data_ptr(),
             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "poco_crop_normalize")
    return out


# outputs of a forward that postprocess() never reads: they stay on the device (uncert_feat alone is 2048-3072 floats per crop)
DEVICE_ONLY_KEYS = ("uncert_feat", "body_feat2", "pred_pose6d", "pred_pose_6d", "pred_cam_t", "pred_fullimg_cam_t", "record",
                    "pred_segm_mask", "backbone_feat")


class POCOTester:
    def __init__(self, args):
        self.args = args
        self.model_cfg = update_hparams(args.cfg)
        # demo.py:305 passes store_false: default True = kinematic post-processing on
        self.model_cfg.POCO.KINEMATIC_UNCERT = getattr(args, "no_kinematic_uncert", True)
        self.backbone = self.model_cfg.POCO.BACKBONE
        # one process per GPU (demo.py --gpus N): LOCAL_RANK picks the device; ranks beyond the visible devices
        # (gloo smoke runs on a smaller box) share them round-robin
        ndev = max(torch.cuda.device_count(), 1)
        self.device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0')) % ndev}")
        torch.cuda.set_device(self.device)
        kw = model_kwargs(self.model_cfg)
        self.model = POCO(**kw, pretrained=args.ckpt, inf_model=getattr(args, "inf_model", "best"),
                          max_batch=max(int(args.batch_size), 1), smpl=args.smpl, device=str(self.device),
                          keep_state_dict=False).finalize()     # (state_dict() re-reads the checkpoint file on demand)
        self.faces = None                       # triangle list for --save_obj, if the body-model file carries one
        if isinstance(args.smpl, str) and os.path.isfile(args.smpl):
            with np.load(args.smpl) as z:
                if "faces" in z.files:
                    self.faces = np.asarray(z["faces"], np.int64)

    def _save_meshes(self, folder: str, verts: np.ndarray, names):
        from .postproc import write_obj
        for v, n in zip(verts, names):
            write_obj(os.path.join(folder, n + ".obj"), v, self.faces)

    # ---- one batch of detections of one frame ---------------------------------------------------
    def make_batch(self, frame_u8: torch.Tensor, dets: np.ndarray, bbox_scale: float = 1.0) -> Dict[str, torch.Tensor]:
        H, W = int(frame_u8.shape[0]), int(frame_u8.shape[1])
        raw = np.asarray(dets)
        # the crop follows the detections' own dtype (float64 tracks stay float64: cv2 sees float32(cx) etc. of THOSE values)
        boxes = torch.from_numpy(np.ascontiguousarray(raw.reshape(-1, 4), np.float64 if raw.dtype == np.float64 else np.float32)).to(self.device)
        dets = raw.astype(np.float32).reshape(-1, 4)
        centers = dets[:, :2]
        scales = np.maximum(dets[:, 2], dets[:, 3]) / 200.0                        # tester.py:196
        info = np.stack([calculate_bbox_info(c, s, (H, W)) for c, s in zip(centers, scales)])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)   # noqa: E731
        return {"img": crop_normalize(frame_u8, boxes, bbox_scale, self.model_cfg.DATASET.IMG_RES),
                "bbox_info": t(info), "focal_length": t(np.full(len(dets), calculate_focal_length(H, W))),
                "scale": t(scales), "center": t(centers), "orig_shape": t(np.tile([[H, W]], (len(dets), 1)))}

    def postprocess(self, output, dets: np.ndarray, W: int, H: int) -> Dict[str, np.ndarray]:
        dets = np.asarray(dets, dtype=np.float32).reshape(-1, 4)
        pred_cam = output["pred_cam"].cpu().numpy()
        res = {"pred_cam": pred_cam, "orig_cam": postproc.convert_crop_cam_to_orig_img(pred_cam, dets, W, H),
               "verts": output["smpl_vertices"].cpu().numpy(), "pose": output["pred_pose"].cpu().numpy(),
               "betas": output["pred_shape"].cpu().numpy(), "joints3d": output["smpl_joints3d"].cpu().numpy(),
               "bboxes": dets}
        j2d = output["smpl_joints2d"].cpu().numpy()
        if "cliff" not in self.backbone:                                             # tester.py:225-230
            j2d = postproc.convert_crop_coords_to_orig_img(dets, j2d, self.model_cfg.DATASET.IMG_RES)
        res["smpl_joints2d"] = np.concatenate([j2d, np.ones((len(dets), 49, 1), np.float32)], -1)
        res["var"], res["var_global"] = postproc.folder_uncert(output["var_pose"], self.backbone,
                                                               self.model_cfg.POCO.KINEMATIC_UNCERT)   # tester.py:242-245
        return res

    @torch.no_grad()
    def iter_frame_results(self, items, bbox_scale: float = 1.0):
        """items: iterable of (frame uint8 [H,W,3] RGB, detections [n,4]); yields one result dict (or None for a frame without
        detections) per item, in order.  The reference runs one forward per image (pocolib/core/tester.py:168-213); crops are
        independent of each other, so here the crops of CONSECUTIVE frames share forwards of up to `batch_size` crops (a photo
        folder with one person per image would otherwise sit in the launch-bound B = 1 regime) and the
        rows are handed back per frame.  At most `batch_size` crops and their frames are held at a time."""
        from collections import deque
        bs = self.model.max_batch
        queue = deque()                    # frames in arrival order: {"dets", "W", "H", "parts", "todo"}
        pieces, batches = [], []           # crops waiting for a forward: (entry, lo, k) + their batch dicts
        npend = 0

        def flush():
            nonlocal npend
            if not pieces:
                return
            batch = batches[0] if len(batches) == 1 else {k: torch.cat([b[k] for b in batches], 0) for k in batches[0]}
            out = self.model(batch, want_segm=False)
            # one D2H per tensor postprocess() reads (not uncert_feat / body_feat2 / pose6d / cam_t: MBs per batch that nothing
            # downstream uses), sliced per frame below
            out = {n: v.cpu() for n, v in out.items() if torch.is_tensor(v) and n not in DEVICE_ONLY_KEYS}
            self.model.check_status()          # (the copies above synchronised the stream) a timed-out in-kernel wait = invalid rows: raise, write nothing
            off = 0
            for e, lo, k in pieces:
                sl = {n: (v[off:off + k] if torch.is_tensor(v) else v) for n, v in out.items()}
                e["parts"].append(self.postprocess(sl, e["dets"][lo:lo + k], e["W"], e["H"]))
                e["todo"] -= k
                off += k
            pieces.clear()
            batches.clear()
            npend = 0

        def drain():
            while queue and queue[0]["todo"] == 0:
                e = queue.popleft()
                yield None if not e["parts"] else {k: np.concatenate([p[k] for p in e["parts"]], 0) for k in e["parts"][0]}

        for frame, dets in items:
            raw = np.asarray(dets)
            raw = raw.reshape(-1, 4) if raw.size else np.zeros((0, 4), np.float32)
            e = {"dets": raw, "W": frame.shape[1], "H": frame.shape[0], "parts": [], "todo": len(raw)}
            queue.append(e)
            if len(raw):
                fr = torch.from_numpy(np.ascontiguousarray(frame)).to(self.device)
                lo = 0
                while lo < len(raw):
                    k = min(bs - npend, len(raw) - lo)
                    batches.append(self.make_batch(fr, raw[lo:lo + k], bbox_scale))
                    pieces.append((e, lo, k))
                    npend += k
                    lo += k
                    if npend == bs:
                        flush()
            yield from drain()
        flush()
        yield from drain()

    def run_on_frames(self, frames, detections: List[np.ndarray], bbox_scale: float = 1.0):
        """frames: iterable of uint8 [H,W,3] RGB arrays; detections[i]: [n_i,4].  Returns the per-frame results."""
        return list(self.iter_frame_results(zip(frames, detections), bbox_scale))

    @torch.no_grad()
    def run_on_video(self, tracking_results: dict, frames, orig_width: int, orig_height: int, bbox_scale: float = 1.0,
                     raw: bool = False):
        """Per-track results like pocolib/core/tester.py:362-479.

        tracking_results: {person_id: {'bbox': [T,4] (cx,cy,w,h), 'frames': [T] frame indices}} - what
        multi_person_tracker hands to the reference (tester.py:113-151).  frames: sequence or callable
        frame index -> uint8 [H,W,3] RGB.

        The reference walks person by person and re-reads every frame for every person
        (dataset/inference.py:72-135).  Here the stream is frame-major: a frame is decoded and uploaded once,
        all people visible in it are cropped on the GPU, crops of consecutive frames are packed into full
        batches of `batch_size`, and the regressed rows are scattered back to the tracks.
        raw=True: return the un-post-processed per-frame network outputs per track (multi-GPU merge, see
        _merge_rank_results)."""
        get = frames if callable(frames) else (lambda i: frames[i])
        bs = self.model.max_batch
        # (frame, person, slot in that person's track), frame-major
        items = sorted((int(f), pid, k) for pid, tr in tracking_results.items() for k, f in enumerate(tr["frames"]))
        keys = ("pred_cam", "smpl_vertices", "pred_pose", "pred_shape", "smpl_joints3d", "smpl_joints2d", "var_pose")
        store = {pid: {k: [None] * len(tr["frames"]) for k in keys} for pid, tr in tracking_results.items()}
        pend_batches, pend_meta = [], []

        def flush():
            if not pend_meta:
                return
            batch = {k: torch.cat([b[k] for b in pend_batches], 0) for k in pend_batches[0]}
            out = self.model(batch, want_segm=False)
            host = {k: out[k].cpu().numpy() for k in keys}
            self.model.check_status()          # after the synchronising copies, before the rows are stored
            for row, (pid, slot) in enumerate(pend_meta):
                for k in keys:
                    store[pid][k][slot] = host[k][row]
            pend_batches.clear()
            pend_meta.clear()

        i = 0
        while i < len(items):
            f = items[i][0]
            j = i
            while j < len(items) and items[j][0] == f:
                j += 1
            fr = torch.from_numpy(np.ascontiguousarray(get(f))).to(self.device, non_blocking=True)
            group = items[i:j]
            while group:                     # a frame with more people than fit goes out in pieces
                room = bs - len(pend_meta)
                part, group = group[:room], group[room:]
                dets = np.stack([np.asarray(tracking_results[pid]["bbox"][slot]) for _, pid, slot in part])   # keeps the tracks' dtype
                pend_batches.append(self.make_batch(fr, dets, bbox_scale))
                pend_meta.extend((pid, slot) for _, pid, slot in part)
                if len(pend_meta) == bs:
                    flush()
            i = j
        flush()

        stacked = {pid: {k: np.stack(v) for k, v in store[pid].items()} for pid in tracking_results}
        if raw:
            return stacked
        return {pid: self._finish_track(tr, stacked[pid], orig_width, orig_height) for pid, tr in tracking_results.items()}

    def _finish_track(self, tr: dict, st: Dict[str, np.ndarray], orig_width: int, orig_height: int,
                      smooth: Optional[bool] = None) -> dict:
        """Per-track post-processing of tester.py:416-479 on the raw per-frame network outputs `st` (keys: pred_cam,
        smpl_vertices, pred_pose, pred_shape, smpl_joints3d, smpl_joints2d, var_pose): optional one-euro smoothing
        (meshes / 3-D joints re-derived from the filtered pose; the 2-D joints stay those of the raw prediction, as in
        the reference), uncertainty post-processing, camera / keypoint conversion to the original image."""
        bboxes = np.asarray(tr["bbox"], np.float32).reshape(-1, 4)
        pose, betas = st["pred_pose"], st["pred_shape"]
        verts, j3d = st["smpl_vertices"], st["smpl_joints3d"]
        if getattr(self.args, "smooth", False) if smooth is None else smooth:               # tester.py:442-447
            from .smooth import smooth_pose
            verts, pose, j3d = smooth_pose(self.model, pose, betas, getattr(self.args, "min_cutoff", 0.004),
                                           getattr(self.args, "beta", 1.5))
        var, var_global = postproc.video_uncert(st["var_pose"], self.backbone,
                                                self.model_cfg.POCO.KINEMATIC_UNCERT)    # tester.py:416-419
        return {
            "pred_cam": st["pred_cam"],
            "orig_cam": postproc.convert_crop_cam_to_orig_img(st["pred_cam"], bboxes, orig_width, orig_height),
            "verts": verts, "pose": pose, "betas": betas, "joints2d": None,
            "smpl_joints3d": j3d,
            "smpl_joints2d": postproc.convert_crop_coords_to_orig_img(bboxes, st["smpl_joints2d"],
                                                                      self.model_cfg.DATASET.IMG_RES),
            "var": var, "var_global": var_global,
            "bboxes": bboxes, "frame_ids": np.asarray(tr["frames"]),
        }

    def run_on_image_folder(self, image_folder: str, detections, output_path: str, bbox_scale=1.0):
        """pocolib/core/tester.py:153-245: images are streamed (decode -> regress -> write), every `--skip_frame`-th image of
        the sorted listing (tester.py:171), so host memory does not grow with the folder.  Unlike the reference's one
        forward per image, consecutive images share forwards of up to --batch_size crops (iter_frame_results); what is
        written per image is the same.  The host side is a pipeline: a small thread pool decodes the next images while the
        GPU regresses (PIL releases the GIL while decoding), another one compresses and writes the per-image .npz files (zlib
        releases it too); at most 2 x batch_size decoded images / unwritten results are in flight."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        from PIL import Image
        names_all = sorted(x for x in os.listdir(image_folder) if x.lower().endswith(IMG_EXT))
        skip = max(int(getattr(self.args, "skip_frame", 1) or 1), 1)
        os.makedirs(output_path, exist_ok=True)
        picked = [(pos, names_all[pos]) for pos in range(0, len(names_all), skip)]
        counts = []
        nthreads = max(1, min(8, (os.cpu_count() or 2) // 2))
        # decode look-ahead / unwritten results in flight: about one batch, and bounded in bytes by the first image's size
        # (2 x batch_size 1080p frames were 0.8 GB at batch_size 64)
        ahead = max(4, self.model.max_batch)
        ahead_bytes = 256 << 20

        def decode(n):
            return np.asarray(Image.open(os.path.join(image_folder, n)).convert("RGB"))

        def dets_of(pos, n, img):
            d = None
            if isinstance(detections, dict):
                d = detections.get(n)
            elif detections is not None and pos < len(detections):      # reference cache: indexed by image position
                d = detections[pos]
            if d is not None and len(d) > 0:
                d = np.asarray(d).reshape(-1, 4)
                return d if d.dtype in (np.float32, np.float64) else d.astype(np.float32)
            H, W = img.shape[:2]                                        # no detector in scope: one centred square box
            s = float(min(H, W))
            return np.array([[W / 2.0, H / 2.0, s, s]], dtype=np.float32)

        def write(n, r):
            np.savez_compressed(os.path.join(output_path, os.path.splitext(n)[0] + "_poco.npz"), **r)
            if getattr(self.args, "save_obj", False):          # tester.py:300-303
                self._save_meshes(os.path.join(output_path, "meshes", os.path.splitext(n)[0]), r["verts"],
                                  [f"{i:06d}" for i in range(len(r["verts"]))])

        t0 = time.time()
        n_img = 0
        with ThreadPoolExecutor(nthreads) as dec_pool, ThreadPoolExecutor(nthreads) as wr_pool:
            def items():
                nonlocal ahead
                q = deque()
                it = iter(picked)
                done = False

                def fill():
                    nonlocal done
                    while not done and len(q) < ahead:
                        nxt = next(it, None)
                        if nxt is None:
                            done = True
                        else:
                            q.append((nxt[0], nxt[1], dec_pool.submit(decode, nxt[1])))

                fill()
                while q:
                    pos, n, fut = q.popleft()
                    img = fut.result()
                    ahead = max(4, min(ahead, ahead_bytes // max(1, img.nbytes)))       # big images: fewer of them decoded ahead
                    fill()
                    d = dets_of(pos, n, img)
                    counts.append(len(d))
                    yield img, d

            writes = deque()
            for (pos, n), r in zip(picked, self.iter_frame_results(items(), bbox_scale)):
                n_img += 1
                if r is not None:
                    writes.append(wr_pool.submit(write, n, r))
                while len(writes) > ahead:
                    writes.popleft().result()
            for w in writes:
                w.result()                                      # re-raises a failed write
        torch.cuda.synchronize()
        dt = time.time() - t0
        n_crops = int(sum(counts))
        return {"images": n_img, "crops": n_crops, "seconds": dt, "fps": n_img / max(dt, 1e-9),
                "crops_per_s": n_crops / max(dt, 1e-9)}


def _run_on_video_folder(self, frame_folder: str, tracking_path: Optional[str], output_path: str, bbox_scale=1.0):
    """demo.py --mode video on a folder of extracted frames (demo.py:60-160): per-track results written as
    <output>/poco_results.npz (the reference joblib-dumps the same dict, demo.py:147-150)."""
    from PIL import Image
    names = sorted(x for x in os.listdir(frame_folder) if x.lower().endswith(IMG_EXT))
    if not names:
        raise FileNotFoundError(f"no frames in {frame_folder}")
    first = np.asarray(Image.open(os.path.join(frame_folder, names[0])).convert("RGB"))
    H, W = first.shape[:2]
    if tracking_path:
        tracking = load_tracking(tracking_path)
    else:
        s = float(min(H, W))
        tracking = {"0": {"bbox": np.tile([[W / 2.0, H / 2.0, s, s]], (len(names), 1)).astype(np.float32),
                          "frames": np.arange(len(names))}}
    skip = max(int(getattr(self.args, "skip_frame", 1)), 1)
    if skip > 1:
        tracking = {k: {"bbox": v["bbox"][::skip], "frames": v["frames"][::skip]} for k, v in tracking.items()}
    load = lambda i: np.asarray(Image.open(os.path.join(frame_folder, names[i])).convert("RGB"))   # noqa: E731
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
    rank = tdist.get_rank() if world > 1 else 0
    t0 = time.time()
    mine = tracking
    if world > 1:
        # whole tracks are the unit of sharding (temporal smoothing stays local: the owning rank smooths and
        # finishes its tracks); the only exchange is one all-gather of the packed per-frame SMPL records, from which rank 0
        # - the writer - re-derives the meshes of the tracks regressed elsewhere
        from . import dist as pdist
        mine = pdist.shard_tracks(tracking, rank, world)
        raw = self.run_on_video(mine, load, W, H, bbox_scale, raw=True) if mine else {}
        results = self._merge_rank_results(raw, tracking, W, H)
    else:
        results = self.run_on_video(tracking, load, W, H, bbox_scale)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if rank != 0:
        tdist.barrier()
        return {"rank": rank, "tracks_local": len(mine), "seconds": dt}
    try:
        os.makedirs(output_path, exist_ok=True)
        flat = {f"{pid}/{k}": v for pid, r in results.items() for k, v in r.items() if v is not None}
        np.savez_compressed(os.path.join(output_path, "poco_results.npz"), **flat)
        if getattr(self.args, "save_obj", False):                      # tester.py:532-535
            for pid, r in results.items():
                sub = f"{int(pid):04d}" if str(pid).lstrip("-").isdigit() else str(pid)
                self._save_meshes(os.path.join(output_path, "meshes", sub), r["verts"], [f"{int(f):06d}" for f in r["frame_ids"]])
    finally:
        if world > 1:               # the other ranks wait here: release them also when the write fails
            tdist.barrier()
    n_crops = sum(len(v["frames"]) for v in tracking.values())
    n_frames = len({int(f) for v in tracking.values() for f in v["frames"]})
    return {"images": n_frames, "crops": n_crops, "tracks": len(tracking), "seconds": dt, "ranks": world,
            "fps": n_frames / max(dt, 1e-9), "crops_per_s": n_crops / max(dt, 1e-9)}


def _merge_rank_results(self, local_raw: dict, tracking: dict, W: int, H: int) -> dict:
    """Multi-GPU video mode.  The OWNING rank finishes its tracks (one-euro smoothing included, _finish_track) and packs
    per frame [final pose 216 | betas 10 | cam 3 | raw var_pose 24 | global confidence 1 | raw crop-space 2-D joints 98];
    ONE all-gather (RCCL when the group's backend is nccl) moves 1.4 KB per crop instead of 83 KB of vertices.  Only rank 0
    - the writer - re-derives the meshes / 3-D joints of the tracks regressed elsewhere from the gathered (already
    smoothed) pose with the batched LBS operator; the other ranks return their own tracks (no rank repeats
    another rank's LBS, smoothing or D2H)."""
    import torch.distributed as tdist
    from . import dist as pdist
    dev = pdist.collective_device()
    rank = tdist.get_rank()
    out, recs = {}, {}
    for pid, st in local_raw.items():
        fin = self._finish_track(tracking[pid], st, W, H)
        out[pid] = fin
        t = {"pred_pose": torch.from_numpy(np.ascontiguousarray(fin["pose"], np.float32)), "pred_shape": torch.from_numpy(st["pred_shape"]),
             "pred_cam": torch.from_numpy(st["pred_cam"]), "var_pose": torch.from_numpy(st["var_pose"])}
        rec = pdist.pack_records(t, head=self.backbone, kinematic=self.model_cfg.POCO.KINEMATIC_UNCERT)
        j2 = torch.from_numpy(np.ascontiguousarray(st["smpl_joints2d"], np.float32)).reshape(rec.shape[0], 98)
        recs[str(pid)] = torch.cat([rec, j2], 1).to(dev)
    full = pdist.gather_track_records(recs, device=dev, width=pdist.VIDEO_REC)
    if rank != 0:
        return out
    for pid, tr in tracking.items():
        if pid in out:
            continue
        rec = full[str(pid)].cpu().numpy()
        T = rec.shape[0]
        pose, betas, cam = rec[:, 0:216].reshape(T, 24, 3, 3), rec[:, 216:226], rec[:, 226:229]
        verts = np.empty((T, 6890, 3), np.float32)
        j3d = np.empty((T, 49, 3), np.float32)
        for lo in range(0, T, self.model.max_batch):
            hi = min(T, lo + self.model.max_batch)
            v, j = self.model.smpl_lbs(torch.from_numpy(np.ascontiguousarray(betas[lo:hi])).to(self.device),
                                       torch.from_numpy(np.ascontiguousarray(pose[lo:hi])).to(self.device))
            verts[lo:hi], j3d[lo:hi] = v.cpu().numpy(), j.cpu().numpy()
        st = {"pred_cam": cam, "smpl_vertices": verts, "pred_pose": pose, "pred_shape": betas, "smpl_joints3d": j3d,
              "smpl_joints2d": rec[:, pdist.REC:pdist.VIDEO_REC].reshape(T, 49, 2), "var_pose": rec[:, 229:253]}
        out[pid] = self._finish_track(tr, st, W, H, smooth=False)        # the owner already smoothed this pose
    return {pid: out[pid] for pid in tracking}


POCOTester.run_on_video_folder = _run_on_video_folder
POCOTester._merge_rank_results = _merge_rank_results


def _load_any(path: str):
    """json, or the joblib/pickle caches the reference writes next to its outputs (demo.py:125-131,163-169)."""
    if path.lower().endswith((".pkl", ".pickle", ".joblib")):
        try:
            import joblib
            return joblib.load(path)
        except ImportError:
            import pickle
            with open(path, "rb") as f:
                return pickle.load(f)
    with open(path) as f:
        return json.load(f)


def load_detections(path: Optional[str]):
    """{image name: [[cx,cy,w,h],...]} (json) or the reference's `detection_results.pkl`: a sequence indexed by the
    position of the image in the sorted folder listing (tester.py:153-169).  None = no detector output."""
    return _load_any(path) if path else None


def load_tracking(path: Optional[str]) -> Optional[dict]:
    """{person_id: {'bbox': [T,4] (cx,cy,w,h), 'frames': [T]}} from json or the reference's
    `tracking_results_<method>.pkl` (demo.py:125-131); tracks shorter than MIN_NUM_FRAMES = 25 frames are dropped as in
    tester.py:133-136."""
    if not path:
        return None
    raw = _load_any(path)
    out = {}
    for k, v in raw.items():
        frames = np.asarray(v["frames"], np.int64).reshape(-1)
        if path.lower().endswith(".json") or frames.shape[0] >= 25:
            out[str(k)] = {"bbox": np.asarray(v["bbox"], np.float32).reshape(-1, 4), "frames": frames}
    return out
