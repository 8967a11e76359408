"""hipGraph forward time of a variant at a batch size with the committed table: python tools/fwd_time.py variant B"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
variant, B = sys.argv[1], int(sys.argv[2])
m = util.make_engine(variant, max_batch=B)
batch = util.cuda_batch(synth.synth_batch(B, 1), torch.device("cuda:0"))
out = m._alloc_outputs(B, False)
for _ in range(8):
    m.graph_forward(batch, out)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40):
        m.graph_forward(batch, out)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 40 * 1e3)
print(f"{variant} B={B}: {min(ts):.4f} ms/forward ({B / min(ts) * 1e3:.0f} crops/s)")
