"""Round 4: ALG 4 with 3-deep rings (cfg.MT = 3) against the table's 2-deep entry, solo and inside the forward.
  python tools/deep_ab.py variant B [--write]"""
import json, re, sys, time
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
    if d is None:
        continue
    c = tuple(m.conv_cfg(i, B))
    if c[6] == 4:
        shapes.setdefault(tune.shape_key(B, *d[:6]), [c, d, []])[2].append(i)
base = fwd_ms()
print(f"{variant} B={B}: table {base:.4f} ms; ALG 4 shapes: {[(k, v[0], len(v[2])) for k, v in shapes.items()]}", flush=True)
cur_t, picked = base, {}
for k, (c, d, idx) in sorted(shapes.items(), key=lambda kv: -len(kv[1][2])):
    H, W, Cin, Cout = d[:4]
    x = torch.randn(B, H, W, Cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    deep = (3,) + c[1:]
    try:
        s2 = min(ops.bench_conv2d(x, w, 1, cfg=c, iters=40)[0] for _ in range(3)) * 1e3
        s3 = min(ops.bench_conv2d(x, w, 1, cfg=deep, iters=40)[0] for _ in range(3)) * 1e3
    except PocoHipError:
        print(f"  {k} x{len(idx)}: {deep} refused"); continue
    for i in idx:
        m.set_conv_cfg(i, B, deep)
    t = fwd_ms()
    keep = t < cur_t * 0.998
    print(f"  {k} x{len(idx)}: solo {s2:.1f} -> {s3:.1f} us; forward {cur_t:.4f} -> {t:.4f} ms {'KEEP' if keep else 'revert'}", flush=True)
    if keep:
        cur_t, picked[k] = t, (deep, s3 * 1e-3)
    else:
        for i in idx:
            m.set_conv_cfg(i, B, c)
print(f"{variant} B={B}: {base:.4f} -> {fwd_ms():.4f} ms, {len(picked)} entries on 3-deep rings")
if "--write" in sys.argv and picked:
    full = json.loads(tune.TABLE.read_text())
    for k, (c, ms) in picked.items():
        ent = full.setdefault(k, {"heuristic_ms": 0.0})
        ent.update({"cfg": list(c), "ms": round(float(ms), 5), "in_context": True, "uses": len(shapes[k][2])})
    tune.TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
    print("wrote", tune.TABLE)
