"""Per-POSITION configurations: does the forward get shorter when the LAST convs of a branch chain (the ones that run while the
other lanes have already reached the module's join) use another configuration than the rest of the chain?
python tools/tail_cfg_probe.py [variant] [B] [blocks]     (blocks: how many trailing BasicBlocks of a chain count as the tail)"""
import re
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth, tune  # noqa: E402
from poco_amd._lib import PocoHipError, lib  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = util.make_engine(variant, max_batch=B)
batch = util.cuda_batch(synth.synth_batch(B, 1), torch.device("cuda:0"))
out = m._alloc_outputs(B, False)
L = lib()


def forward_ms(iters=25):
    for _ in range(3):
        m(batch, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m(batch, out=out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


names = [o[0] for o in m.ops()]
base = forward_ms()
print(f"{variant} B={B}: {base:.3f} ms/forward")
for br in range(4):
    blocks = "".join(str(3 - k) for k in range(nblk))
    pat = re.compile(r"stage[234]\.\d\.branches\.%d\.[%s]\.conv[12]$" % (br, blocks))
    idxs = [i for i, n in enumerate(names) if pat.search(n) and m.conv_desc(i) is not None]
    if not idxs:
        continue
    H, W, Cin, Cout, ks, st = m.conv_desc(idxs[0])[:6]
    solo = sorted(r for r in tune.solo_times(L, B, H, W, Cin, Cout, ks, st) if r[0] > 0)[:10]
    cur = tuple(m.conv_cfg(idxs[0], B))
    best, best_t = cur, forward_ms()
    print(f"branch {br}: {len(idxs)} tail convs {H}x{W} {Cin}->{Cout}, current {cur}: {best_t:.3f} ms")
    for ms, cfg in solo:
        if tuple(cfg) == cur:
            continue
        try:
            for i in idxs:
                m.set_conv_cfg(i, B, cfg)
        except PocoHipError:
            continue
        t = forward_ms()
        print(f"     {tuple(cfg)} solo {ms*1e3:6.1f} us -> forward {t:.3f} ms")
        if t < best_t - 0.01:
            best, best_t = tuple(cfg), t
    for i in idxs:
        m.set_conv_cfg(i, B, best)
    print(f"  kept {best}: {best_t:.3f} ms")
print(f"after: {forward_ms():.3f} ms/forward (was {base:.3f})")
