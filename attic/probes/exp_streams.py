"""Experiment: does running independent work on several HIP streams hide per-kernel fixed costs?
Two engines (B each) on 1 stream sequentially vs on 2 streams concurrently."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NE = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
engines = [util.make_engine(variant, max_batch=B) for _ in range(NE)]
batches = [util.cuda_batch(synth.synth_batch(B, 10 + i), dev) for i in range(NE)]
outs = [e._alloc_outputs(B, False) for e in engines]
streams = [torch.cuda.Stream() for _ in range(NE)]


def run(concurrent, iters=10):
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        for i, e in enumerate(engines):
            s = streams[i] if concurrent else streams[0]
            with torch.cuda.stream(s):
                e(batches[i], out=outs[i])
    torch.cuda.synchronize()
    return (time.time() - t0) / iters


for c in (False, True):
    run(c, 3)
for c in (False, True, False, True):
    dt = run(c)
    print(f"{variant} {NE}x B={B} {'concurrent streams' if c else 'one stream       '}: {dt*1e3:.2f} ms -> {NE*B/dt:.0f} crops/s")
