import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from poco_amd import synth
from tests import util
for variant in ("hrnet_w48_cls-cliff", "resnet50-cliff", "hrnet_w32-pare"):
    for B in (1, 3):
        m = util.make_engine(variant, max_batch=B)
        bnp = synth.synth_batch(B, 5 + B)
        out = m(util.cuda_batch(bnp, torch.device("cuda:0")))
        ref = util.oracle_forward(variant, bnp)
        worst = 0.0
        for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
            worst = max(worst, float((out[k].cpu() - ref[k]).abs().max()))
        print(variant, "B =", B, "max abs dev vs oracle %.2e" % worst)
        assert worst < 1e-3
