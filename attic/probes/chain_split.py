"""Per-op configuration inside a branch chain: the convs of one shape (e.g. the 24 7x7 384->384 convs = 3 modules x 8) get
configuration A while the other branches still run and configuration B for the last (8 - k) convs of every module, when the chain runs
alone.  python tools/chain_split.py <shape key> "<cfg B as 7 ints>" [period]   (cfg A = the table's)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd import synth, tune  # noqa: E402
from tests import util  # noqa: E402
key = sys.argv[1]
cfgB = tuple(int(x) for x in sys.argv[2].split(","))
period = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 4
B = int(key.split("x")[0])
m = util.make_engine("hrnet_w48_cls-cliff", max_batch=B)
batch = util.cuda_batch(synth.synth_batch(B, 1), torch.device("cuda:0"))
out = m._alloc_outputs(B, False)
m.set_num_lanes(lanes)
idxs = [i for i in range(len(m.ops())) if m.conv_desc(i) is not None and tune.shape_key(B, *m.conv_desc(i)[:6]) == key]
cfgA = tuple(m.conv_cfg(idxs[0], B))
print(len(idxs), "ops of", key, "A =", cfgA, "B =", cfgB)
def fwd_ms(iters=30):
    for _ in range(5):
        m(batch, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        m(batch, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
for k in list(range(period, -1, -1)):
    for n, i in enumerate(idxs):
        m.set_conv_cfg(i, B, cfgB if (n % period) >= k else cfgA)
    print(f"first {k} of every {period} on A, rest on B: {fwd_ms():.3f} ms", flush=True)
