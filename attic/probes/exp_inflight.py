"""Throughput with k forwards in flight: k engines (own weights + workspace), k streams, hipGraph replay round-robin."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
for k in (1, 2, 3):
    ms = [util.make_engine(variant, max_batch=B) for _ in range(k)]
    batches = [util.cuda_batch(synth.synth_batch(B, 1 + i), dev) for i in range(k)]
    outs = [m._alloc_outputs(B, want_segm=False) for m in ms]
    streams = [torch.cuda.Stream() for _ in range(k)]
    for i in range(k):
        with torch.cuda.stream(streams[i]):
            for _ in range(3):
                ms[i].graph_forward(batches[i], outs[i])
    torch.cuda.synchronize()
    steps = 30
    t0 = time.perf_counter()
    for s in range(steps):
        with torch.cuda.stream(streams[s % k]):
            ms[s % k].graph_forward(batches[s % k], outs[s % k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{variant} B={B} in flight {k}: {steps * B / dt:.1f} crops/s ({dt / steps * 1e3:.2f} ms per forward)")
    del ms, outs, batches
