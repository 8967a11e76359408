"""In-forward per-op time (HIP events, single lane) and hipGraph forward time of one 1x1 shape under several configurations:
    python tools/shape_cfg_ops.py variant B HxWxCinxCout ks "cfg" ["cfg" ...]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
variant, B = sys.argv[1], int(sys.argv[2])
H, W, Cin, Cout = map(int, sys.argv[3].split("x"))
ks = int(sys.argv[4])
cfgs = [tuple(int(x) for x in c.split(",")) for c in sys.argv[5:]]
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine(variant, max_batch=B)
m(batch)
idxs = [i for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None and tuple(m.conv_desc(i)[:5]) == (H, W, Cin, Cout, ks)]
cfgs = [tuple(m.conv_cfg(idxs[0], B))] + cfgs
out = m._alloc_outputs(B, False)
for c in cfgs:
    for i in idxs:
        m.set_conv_cfg(i, B, c)
    m.set_num_lanes(1)
    for _ in range(2):
        m(batch)
    prof = m.profile_ops(batch, iters=8)
    us = sum(prof[i][3] for i in idxs) / len(idxs) * 1e3
    m.set_num_lanes(4)
    m.release_graphs()
    for _ in range(6):
        m.graph_forward(batch, out)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            m.graph_forward(batch, out)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
    print(f"{H}x{W} {Cin}->{Cout} x{len(idxs)} {c}: {us:.1f} us per op in the forward, forward {best:.4f} ms", flush=True)
