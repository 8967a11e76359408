import sys
sys.path.insert(0, '/root/repo')
import torch
from poco_amd import synth
from tests import util
import os
V = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("VARIANT", "hrnet_w48_cls-cliff")
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
m = util.make_engine(V, max_batch=NB)
m.set_num_lanes(1)
batch = util.cuda_batch(synth.synth_batch(NB, 1), torch.device("cuda:0"))
for _ in range(2): m(batch)
prof = m.profile_ops(batch, iters=5)
for i, (nm, fl, ty, ms) in enumerate(prof):
    if os.environ.get("ALL") or len(sys.argv) > 2 or i < 34 or i > len(prof) - 45:
        d = m.conv_desc(i)
        cfg = m.conv_cfg(i, NB) if d is not None else None
        tf = fl * NB / (ms * 1e-3) / 1e12 if ms > 0 else 0
        print(f"{i:3d} {nm:44s} ty={ty} {ms*1e3:7.1f} us {tf:6.1f} TF {d[:6] if d else ''} {cfg if cfg else ''}")
