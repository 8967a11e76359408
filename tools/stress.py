"""Race screen: many forwards on the same inputs must be BITWISE identical (LDS-DMA / barrier / lane-join races
show up as rare differing tiles).  usage: python tools/stress.py [variant] [B] [iters]"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
m = util.make_engine(variant, max_batch=B)
batch = util.cuda_batch(synth.synth_batch(B, 5), torch.device("cuda:0"))
keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "uncert_feat")
ref = {k: v.clone() for k, v in m(batch).items() if k in keys}
bad = 0
for mode in ("eager-4lanes", "eager-1lane", "graph"):
    m.set_num_lanes(1 if mode == "eager-1lane" else 4)
    out = m._alloc_outputs(B, False)
    for i in range(iters):
        o = m.graph_forward(batch, out) if mode == "graph" else m(batch)
        if i % 10 == 9 or mode == "graph":
            for k in keys:
                if not torch.equal(o[k], ref[k]):
                    bad += 1
                    print(f"MISMATCH mode={mode} iter={i} key={k} max|d|={float((o[k]-ref[k]).abs().max()):.3e}")
    torch.cuda.synchronize()
    print(mode, "done")
print("stress:", "FAILED" if bad else "OK", f"({3*iters} forwards, {bad} mismatches)")
