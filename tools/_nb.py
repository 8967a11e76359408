import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth
from tests import util
B = 64
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine("resnet50-cliff", max_batch=B)
m(batch)
idxs = [i for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None and tuple(m.conv_desc(i)[:5]) == (14, 14, 256, 256, 3)]
names = [o[0] for o in m.ops()]
for c in [(1, 2, 2, 4, 32, 0, 8), (1, 2, 2, 1, 16, 2, 13), (2, 2, 2, 1, 16, 2, 13), (1, 2, 2, 1, 16, 0, 13), (1, 2, 2, 4, 32, 0, 8)]:
    try:
        for i in idxs:
            m.set_conv_cfg(i, B, c)
    except Exception as e:
        print(c, "invalid", e); continue
    m.set_num_lanes(1)
    for _ in range(2):
        m(batch)
    prof = m.profile_ops(batch, iters=8)
    tot = sum(p[3] for p in prof) * 1e3
    i = idxs[1]
    print(c, "total per-op sum %.1f us;" % tot, "  ".join(f"{names[j].split('backbone.')[-1]} {prof[j][3] * 1e3:.1f}" for j in range(i - 1, i + 3)), flush=True)
