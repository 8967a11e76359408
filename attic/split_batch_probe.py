"""One batch as N sub-batches on N engines / streams against one engine: python tools/split_batch_probe.py variant B nsplit"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
variant, B, NS = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")


def timed(fn, n=40, rounds=3):
    ts = []
    for _ in range(rounds):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / n * 1e3)
    return min(ts)


m = util.make_engine(variant, max_batch=B)
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
out = m._alloc_outputs(B, False)
for _ in range(8):
    m.graph_forward(batch, out)
t1 = timed(lambda: m.graph_forward(batch, out))
print(f"{variant} one engine B={B}: {t1:.4f} ms ({B / t1 * 1e3:.0f} crops/s)")
del m
b = B // NS
engines = [util.make_engine(variant, max_batch=b) for _ in range(NS)]
batches = [util.cuda_batch(synth.synth_batch(b, 1 + i), dev) for i in range(NS)]
outs = [e._alloc_outputs(b, False) for e in engines]
streams = [torch.cuda.Stream() for _ in range(NS)]
for e, bt, o, s in zip(engines, batches, outs, streams):
    with torch.cuda.stream(s):
        for _ in range(8):
            e.graph_forward(bt, o)
torch.cuda.synchronize()


def step():
    cur = torch.cuda.current_stream()
    for e, bt, o, s in zip(engines, batches, outs, streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            e.graph_forward(bt, o)
    for s in streams:
        cur.wait_stream(s)


t2 = timed(step)
print(f"{variant} {NS} engines x B={b}: {t2:.4f} ms ({B / t2 * 1e3:.0f} crops/s)  {100 * (t1 / t2 - 1):+.1f} %")
