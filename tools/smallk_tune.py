"""Round 4: the split-K direct 3x3 conv (ALG 5, ks = 3) against the table's entry for every 3x3 conv shape of a variant at a small
batch, solo and then inside the forward (graph replay).  python tools/smallk_tune.py variant B [--write]"""
import json, sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import ops, synth, tune  # noqa: E402
from poco_amd._lib import PocoHipError  # noqa: E402
from tests import util  # noqa: E402

variant, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine(variant, max_batch=B)
m(batch)


def fwd_ms(reps=60):
    m.release_graphs()
    out = m._alloc_outputs(B, False)
    for _ in range(8):
        m.graph_forward(batch, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.graph_forward(batch, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


shapes = {}
for i, _ in enumerate(m.ops()):
    d = m.conv_desc(i)
    if d is None or d[4] != 3 or d[0] * d[1] <= 1:
        continue
    shapes.setdefault(tune.shape_key(B, *d[:6]), [tuple(m.conv_cfg(i, B)), d, []])[2].append(i)
base = fwd_ms()
print(f"{variant} B={B}: table {base:.4f} ms, {len(shapes)} 3x3 shapes", flush=True)
cur_t, picked = base, {}
for k, (c, d, idx) in sorted(shapes.items(), key=lambda kv: -len(kv[1][2])):
    H, W, Cin, Cout, ks, st = d[:6]
    x = torch.randn(B, H, W, Cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    s0 = min(ops.bench_conv2d(x, w, st, cfg=c, iters=40)[0] for _ in range(3)) * 1e3
    solo = []
    for wm in (4, 8, 16):
        cfg = (1, 1, wm, 1, 1, 1, 5)
        try:
            solo.append((min(ops.bench_conv2d(x, w, st, cfg=cfg, iters=40)[0] for _ in range(3)) * 1e3, cfg))
        except PocoHipError:
            pass
    solo.sort()
    msg = f"  {k} x{len(idx)} {c}: solo {s0:.1f} us | split-K " + " ".join(f"wm{cf[2]}={t:.1f}" for t, cf in solo)
    if not solo or solo[0][0] > 0.92 * s0:
        print(msg, flush=True); continue
    best = solo[0][1]
    try:
        for i in idx:
            m.set_conv_cfg(i, B, best)
    except PocoHipError as e:
        print(msg, "| refused in the engine:", str(e)[:60], flush=True); continue
    t = fwd_ms()
    keep = t < cur_t * 0.999
    print(msg + f" | forward {cur_t:.4f} -> {t:.4f} ms {'KEEP' if keep else 'revert'}", flush=True)
    if keep:
        cur_t, picked[k] = t, (best, solo[0][0] * 1e-3)
    else:
        for i in idx:
            m.set_conv_cfg(i, B, c)
final = fwd_ms()
print(f"{variant} B={B}: {base:.4f} -> {final:.4f} ms ({(final / base - 1) * 100:+.1f} %), {len(picked)} shapes on the split-K conv")
if "--write" in sys.argv and picked and final < base * 0.995:
    full = json.loads(tune.TABLE.read_text())
    for k, (c, ms) in picked.items():
        ent = full.setdefault(k, {"heuristic_ms": 0.0})
        ent.update({"cfg": list(c), "ms": round(float(ms), 5), "in_context": True, "uses": len(shapes[k][2])})
    tune.TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
    print("wrote", tune.TABLE)
