"""End-to-end streaming rate (BASELINE.json config #5 shape: frames -> detections -> crops -> POCO-CLIFF, bs=128)
on synthetic 1080p frames: frame upload over PCIe + GPU crop/normalise + forward + record download.
Detector/tracker are out of scope: boxes are synthetic (PEOPLE per frame).  Prints one JSON line."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd.stream import CropStream  # noqa: E402
from tests import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="hrnet_w48_cls-cliff")
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--people", type=int, default=4)
ap.add_argument("--batches", type=int, default=30)
ap.add_argument("--hw", type=int, nargs=2, default=[1080, 1920])
args = ap.parse_args()
H, W = args.hw
m = util.make_engine(args.variant, max_batch=args.batch)
fpb = args.batch // args.people                       # frames per batch
cs = CropStream(m, (H, W), args.batch, ring=2 * fpb)
rng = np.random.default_rng(0)
frames = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(4)]
boxes = np.stack([np.array([rng.uniform(0.2, 0.8) * W, rng.uniform(0.3, 0.7) * H, s, s], np.float32)
                  for s in rng.uniform(150, 600, args.people)])


def one_batch(i, buf):
    groups = [(cs.upload(frames[(i * fpb + k) % len(frames)]), boxes) for k in range(fpb)]
    return cs.run(groups, buf)


for i in range(3):
    one_batch(i, 0)
torch.cuda.synchronize()
t0 = time.time()
done = None
for i in range(args.batches):
    h, n = one_batch(i, i & 1)
    ev = torch.cuda.Event()
    ev.record()
    if done is not None:
        done[0].synchronize()                         # consume the previous batch's records while this one runs
        _ = float(done[1][0, 0])
    done = (ev, h)
torch.cuda.synchronize()
dt = time.time() - t0
crops = args.batches * fpb * args.people
print(json.dumps({"workload": f"{args.variant} streaming, {H}x{W} uint8 frames, {args.people} people/frame, bs={args.batch}",
                  "frames_per_s": round(args.batches * fpb / dt, 1), "crops_per_s": round(crops / dt, 1),
                  "ms_per_batch": round(dt / args.batches * 1e3, 2), "pcie_in_MB_per_batch": round(fpb * H * W * 3 / 1e6, 1),
                  "includes": "frame H2D (pinned, copy stream), GPU crop+normalise, forward (hipGraph), 253-float record D2H"}))
