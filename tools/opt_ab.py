"""Same-box A/B of engine build options in the hipGraph forward: python tools/opt_ab.py variant B "k=v,k=v" ["k=v" ...]  ("" = defaults)"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
variant, B = sys.argv[1], int(sys.argv[2])
opts = sys.argv[3:] or [""]
batch = util.cuda_batch(synth.synth_batch(B, 1), torch.device("cuda:0"))
parse = lambda o: dict(kv.split("=") for kv in o.split(",") if kv)
ms = [(o or "defaults", util.make_engine(variant, max_batch=B, options=parse(o) or None)) for o in opts]
outs = [m._alloc_outputs(B, False) for _, m in ms]
for (_, m), o in zip(ms, outs):
    for _ in range(8):
        m.graph_forward(batch, o)
best = {n: 1e9 for n, _ in ms}
for _ in range(4):
    for (n, m), o in zip(ms, outs):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            m.graph_forward(batch, o)
        torch.cuda.synchronize(); best[n] = min(best[n], (time.perf_counter() - t0) / 30 * 1e3)
base = best[ms[0][0]]
print(f"{variant} B={B}: " + "  ".join(f"[{n}] {t:.4f} ms ({100 * (base / t - 1):+.2f} %)" for n, t in best.items()))
