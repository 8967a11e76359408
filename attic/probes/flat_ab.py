"""Round 4: FLAT items of ALG 8 (cfg R = 4, NI = 0) against the table's rectangular items.
  python tools/flat_ab.py [variant] [B]
1. solo: poco_bench_conv2d of every ALG 8 shape of the variant at batch B, table cfg vs flat cfg (same MT / NT, and MT 1 / 2);
2. in context: hipGraph forward with (a) the table, (b) every ALG 8 shape whose strip fits on flat items, (c) one shape at a time."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth, tune, ops  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)


def fwd_ms(m, reps=40):
    out = m._alloc_outputs(B, False)
    for _ in range(6):
        m.graph_forward(batch, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.graph_forward(batch, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


m = util.make_engine(variant, max_batch=B)
m(batch)
shapes = {}
for i, _ in enumerate(m.ops()):
    d = m.conv_desc(i)
    if d is None:
        continue
    c = m.conv_cfg(i, B)
    if c[6] == 8:
        shapes.setdefault((d[0], d[1], d[2], d[3]), [c, []])[1].append(i)
print(f"{variant} B={B}: ALG 8 shapes", {k: (v[0], len(v[1])) for k, v in shapes.items()})

flat_ok = {}
for (H, W, Cin, Cout), (c, idx) in shapes.items():
    x = torch.randn(B, H, W, Cin, device=dev)      # (bench_conv2d takes NHWC like ops.conv2d_nhwc)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    row = []
    for cfg in (c, (c[0], c[1], 2, 4, 4, 0, 8), (1, c[1], 2, 4, 4, 0, 8), (2, c[1], 2, 4, 4, 0, 8), (1, 2, 2, 4, 4, 0, 8)):
        try:
            ms = ops.bench_conv2d(x, w, 1, cfg=cfg, iters=30)[0]
            row.append(f"{cfg}: {ms*1e3:.1f} us")
            if cfg[5] == 0 and cfg[1] == c[1] and cfg[0] == c[0]:
                flat_ok[(H, W, Cin, Cout)] = cfg
        except Exception as e:      # strip does not fit at this NT
            row.append(f"{cfg}: refused")
    print(f"solo {H}x{W} {Cin}->{Cout} x{len(idx)}:", " | ".join(row), flush=True)

base = fwd_ms(m)
print(f"in context: table {base:.3f} ms", flush=True)
for key, cfg in flat_ok.items():
    for i in shapes[key][1]:
        m.set_conv_cfg(i, B, cfg)
    t = fwd_ms(m)
    print(f"  + {key} flat {cfg}: {t:.3f} ms ({(t/base-1)*100:+.2f} % vs table)", flush=True)
    m.release_graphs()
allflat = fwd_ms(m)
print(f"in context: all flat {allflat:.3f} ms ({(allflat/base-1)*100:+.2f} %)")
for key in flat_ok:        # back to the table, one at a time from the all-flat state
    for i in shapes[key][1]:
        m.set_conv_cfg(i, B, shapes[key][0])
    t = fwd_ms(m)
    print(f"  all flat except {key}: {t:.3f} ms")
    for i in shapes[key][1]:
        m.set_conv_cfg(i, B, flat_ok[key])
    m.release_graphs()
