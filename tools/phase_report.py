"""Single-stream time per network part (regex buckets over op names)."""
import argparse, re, sys
from collections import defaultdict
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="hrnet_w48_cls-cliff")
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--dump", action="store_true")
args = ap.parse_args()
m = util.make_engine(args.variant, max_batch=args.batch)
m.set_num_lanes(1)
batch = util.cuda_batch(synth.synth_batch(args.batch, 1), torch.device("cuda:0"))
for _ in range(2):
    m(batch)
prof = m.profile_ops(batch, iters=5)
BUCKETS = [("fuse", r"fuse_layers|\.fuse|upsample|bilinear|sum"), ("branches", r"branches"), ("transition", r"transition"),
           ("layer1", r"layer1"), ("stem", r"conv1|conv2|stem|bn1|bn2"), ("cls_head", r"incre|downsamp|final_layer|classifier"),
           ("head", r"head|fc|dec|smpl|cam|uncert|rot6d")]
agg = defaultdict(lambda: [0, 0.0])
tot = 0.0
for nm, fl, ty, ms in prof:
    tot += ms
    for b, rx in BUCKETS:
        if re.search(rx, nm):
            break
    else:
        b = "other"
    agg[b][0] += 1; agg[b][1] += ms
    if args.dump:
        print(f"{nm:60s} ty={ty} {ms*1e3:8.1f} us")
print(f"total {tot:.2f} ms")
for b, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{b:12s} n={n:4d} {ms:7.3f} ms {100*ms/tot:5.1f}%")
