"""Experiment: hipGraph replay of one forward (torch.cuda.CUDAGraph) vs eager launches."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
m = util.make_engine(variant, max_batch=B)
m.set_num_lanes(lanes)
batch = util.cuda_batch(synth.synth_batch(B, 10), dev)
out = m._alloc_outputs(B, False)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        m(batch, out=out)
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in out.items()}


def bench(fn, iters=20):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


with torch.cuda.stream(s):
    t_eager = bench(lambda: m(batch, out=out))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    m(batch, out=out)
for v in out.values():
    v.zero_()
g.replay()
torch.cuda.synchronize()
ok = all(torch.equal(ref[k], out[k]) for k in ref)
t_graph = bench(g.replay)
print(f"{variant} B={B} lanes={lanes}: eager {t_eager*1e3:.2f} ms ({B/t_eager:.0f} crops/s)  graph {t_graph*1e3:.2f} ms ({B/t_graph:.0f} crops/s)  identical={ok}")
