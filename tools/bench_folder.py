"""Folder mode end to end on a synthetic photo folder with ONE person per image: the reference's loop runs one
forward per image (pocolib/core/tester.py:168-213); POCOTester.iter_frame_results lets consecutive images share forwards.

    python tools/bench_folder.py [--images 256] [--batch 64] [--variant hrnet_w48_cls-cliff]
Prints images/s for batch_size 1 (= one forward per image) and for --batch, for the regress-only path (frames already decoded) and
for run_on_image_folder (PNG decode + regress + per-image npz)."""
import argparse
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=256)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--variant", default="hrnet_w48_cls-cliff")
args = ap.parse_args()
CFG = {"hrnet_w48_cls-cliff": "configs/demo_poco_cliff.yaml", "resnet50-cliff": "configs/demo_poco_cliff_resnet50.yaml",
       "hrnet_w32-pare": "configs/demo_poco_pare.yaml"}[args.variant]

import demo  # noqa: E402
from PIL import Image  # noqa: E402
from poco_amd.tester import POCOTester  # noqa: E402

tmp = Path(tempfile.mkdtemp(prefix="poco_folder_"))
w = util.synth_weights(args.variant)
torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in w.items()}}, tmp / "ckpt.pt")
np.savez(tmp / "smpl.npz", **synth.synth_smpl(7))
imgs = tmp / "imgs"
imgs.mkdir()
r = np.random.default_rng(0)
frames, dets = [], {}
for i in range(args.images):
    f = r.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    Image.fromarray(f).save(imgs / f"im{i:05d}.png")
    frames.append(f)
    dets[f"im{i:05d}.png"] = np.array([[r.uniform(200, 440), r.uniform(150, 330), 220, 300]], np.float32)
res = {}
for bs in (1, args.batch):
    a = demo.parse_args(["--cfg", CFG, "--ckpt", str(tmp / "ckpt.pt"), "--mode", "folder", "--image_folder", str(imgs),
                         "--output_folder", str(tmp / f"out{bs}"), "--batch_size", str(bs), "--smpl", str(tmp / "smpl.npz"), "--no_render"])
    t = POCOTester(a)
    dl = [dets[k] for k in sorted(dets)]
    t.run_on_frames(frames[:8], dl[:8])                     # warm-up (tuned table, allocator)
    torch.cuda.synchronize()
    t0 = time.time()
    out = t.run_on_frames(frames, dl)
    torch.cuda.synchronize()
    t1 = time.time() - t0
    st = t.run_on_image_folder(str(imgs), dets, str(tmp / f"out{bs}"))
    res[bs] = (args.images / t1, st["fps"], out)
    print(f"batch_size {bs:3d}: regress-only {args.images / t1:8.1f} images/s | folder end to end (PNG decode + npz write) {st['fps']:8.1f} images/s")
    del t
a, b = res[1][2], res[args.batch][2]
dev = max(float(np.abs(x["pose"] - y["pose"]).max()) for x, y in zip(a, b))
print(f"speed-up regress-only {res[args.batch][0] / res[1][0]:.2f}x, end to end {res[args.batch][1] / res[1][1]:.2f}x; "
      f"max |pose(batched) - pose(per image)| = {dev:.2e}")
