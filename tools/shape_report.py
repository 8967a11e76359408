"""Per conv-shape time report of one variant (single stream, per-op HIP events)."""
import argparse
import sys
from collections import defaultdict
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="hrnet_w48_cls-cliff")
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
m = util.make_engine(args.variant, max_batch=args.batch)
m.set_num_lanes(1)
batch = util.cuda_batch(synth.synth_batch(args.batch, 1), torch.device("cuda:0"))
for _ in range(2):
    m(batch)
prof = m.profile_ops(batch, iters=5)
names = [o[0] for o in m.ops()]
agg = defaultdict(lambda: [0, 0.0, 0.0])
tot = 0.0
for i, (nm, fl, ty, ms) in enumerate(prof):
    tot += ms
    key = ("conv",) + m.conv_desc(i)[:6] if ty == 1 else ("other", ty)
    a = agg[key]
    a[0] += 1; a[1] += ms; a[2] += fl * args.batch
print(f"total {tot:.2f} ms")
cum = 0.0
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    cum += ms
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    ideal = fl / 157.3e12 * 1e3
    print(f"{str(key):44s} n={n:3d} {ms:7.3f} ms ({100*ms/tot:4.1f}% cum {100*cum/tot:4.1f}%) {tf:6.1f} TF  ideal {ideal:6.3f} ms  lost {ms-ideal:6.3f}")
