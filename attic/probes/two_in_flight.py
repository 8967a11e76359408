"""Experiment: TWO batches in flight - two engines (own workspace and side lanes each) replay their hipGraphs on two streams, so
the serial parts of one forward (stem / layer1, cls head, SMPL tail: one kernel resident) overlap the other's multi-lane parts.
Throughput only: the latency of a batch roughly doubles.  python tools/two_in_flight.py [variant] [B] [steps]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda:0")
eng = [util.make_engine(variant, max_batch=B) for _ in range(2)]
batches = [util.cuda_batch(synth.synth_batch(B, 11 + k), dev) for k in range(2)]
outs = [e._alloc_outputs(B, False) for e in eng]
streams = [torch.cuda.Stream() for _ in range(2)]
for k in range(2):
    with torch.cuda.stream(streams[k]):
        for _ in range(3):
            eng[k].graph_forward(batches[k], outs[k])
torch.cuda.synchronize()


def run(n_streams, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            eng[k].graph_forward(batches[k], outs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    one = run(1, steps)
    two = run(2, steps)
    print(f"{variant} B={B}: one in flight {one:.3f} ms/step = {B / one * 1e3:.0f} crops/s | two in flight {two:.3f} ms/step = "
          f"{B / two * 1e3:.0f} crops/s ({(one / two - 1) * 100:+.1f} %)")
# results must not depend on the overlap
ref = {k: v.clone() for k, v in outs[0].items() if isinstance(v, torch.Tensor)}
run(2, 10)
for k, v in ref.items():
    assert torch.equal(v, outs[0][k]), k
print("outputs bitwise unchanged under overlap")
