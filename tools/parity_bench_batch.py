import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from poco_amd import synth
from tests import util
import os
CASES = [(v, int(b)) for v, b in (c.split(":") for c in os.environ.get("CASES", "hrnet_w48_cls-cliff:64,hrnet_w32-pare:32").split(","))]
for variant, B in CASES:
    m = util.make_engine(variant, max_batch=B)
    bnp = synth.synth_batch(B, 1234)
    out = m(util.cuda_batch(bnp, torch.device("cuda:0")))
    n7 = sum(1 for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None and m.conv_cfg(i, B)[6] == 7)
    ref = util.oracle_forward(variant, bnp)
    devs = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices")}
    print(variant, "B =", B, "convs on ALG 7:", n7, {k: "%.2e" % v for k, v in devs.items()})
