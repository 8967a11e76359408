"""Quick end-to-end check on the GPU box: run a variant, compare with the oracle, print per-op timing."""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="hrnet_w48_cls-cliff")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--check", type=int, default=2)
ap.add_argument("--top", type=int, default=25)
ap.add_argument("--lanes", type=int, default=4)
args = ap.parse_args()
dev = torch.device("cuda:0")
t0 = time.time()
m = util.make_engine(args.variant, max_batch=args.batch)
m.set_num_lanes(args.lanes)
print(f"engine ready in {time.time()-t0:.1f}s, workspace {m.workspace_bytes()/1e9:.2f} GB, ops {len(m.ops())}")
if args.check:
    bnp = synth.synth_batch(args.check, 1234)
    ref = util.oracle_forward(args.variant, bnp)
    out = m(util.cuda_batch(bnp, dev))
    torch.cuda.synchronize()
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "smpl_joints3d"):
        print(f"  {k:16s} max|d| = {np.abs(out[k].cpu().numpy() - ref[k].numpy()).max():.3e}")
batch = util.cuda_batch(synth.synth_batch(args.batch, 1), dev)
for _ in range(3):
    m(batch)
torch.cuda.synchronize()
t0 = time.time()
n = 10
for _ in range(n):
    m(batch)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
gf = sum(f for _, f, _ in m.ops()) * args.batch / 1e9
print(f"forward B={args.batch}: {dt*1e3:.2f} ms  -> {args.batch/dt:.1f} crops/s, {gf/dt/1e3:.1f} TFLOP/s ({gf/dt/1e3/157.3:.3f} of fp32 MFMA peak)")
prof = m.profile_ops(batch, iters=3)
tot = sum(p[3] for p in prof)
print(f"sum of per-op times {tot:.2f} ms")
agg = {}
for nm, fl, ty, ms in prof:
    agg.setdefault(ty, [0, 0.0, 0.0])
    agg[ty][0] += 1; agg[ty][1] += ms; agg[ty][2] += fl
names = ["STEM", "CONV", "MAXPOOL", "BILINEAR", "FUSE", "AVGPOOL", "ATTN", "LC2D", "ROT6D", "COPY", "BCAST", "SMPL", "CAMERA", "NCHW_OUT"]
for ty, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {names[ty]:9s} n={cnt:4d} {ms:8.3f} ms  {fl*args.batch/1e9/ max(ms,1e-9):8.1f} GFLOP/ms")
for nm, fl, ty, ms in sorted(prof, key=lambda p: -p[3])[:args.top]:
    d = m.conv_desc([o[0] for o in m.ops()].index(nm)) if ty == 1 else None
    tf = fl * args.batch / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"  {ms:7.3f} ms {tf:6.1f} TF  {nm}  {d}")
