"""Round 4: ALG 11 (F(4x4) as 36 position GEMMs, three launches, chained) on the 14x14 planes INSIDE the forward, on top of whatever
the table says for the other shapes.  python tools/wg14_ab.py [variant] [B]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "hrnet_w48_cls-cliff"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)


def fwd_ms(m, reps=40):
    m.release_graphs()
    out = m._alloc_outputs(B, False)
    for _ in range(6):
        m.graph_forward(batch, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.graph_forward(batch, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


m = util.make_engine(variant, max_batch=B, options={"wg_max_plane": 16})
ref = m(batch)
ref = {k: v.clone() for k, v in ref.items() if torch.is_tensor(v)}
idx = [i for i, _ in enumerate(m.ops()) if (d := m.conv_desc(i)) is not None and d[4] == 3 and d[5] == 1 and d[0] == 14 and d[1] == 14]
cur = {i: m.conv_cfg(i, B) for i in idx}
base = fwd_ms(m)
print(f"{variant} B={B}: {len(idx)} 14x14 convs, table {base:.3f} ms", flush=True)
for cfg in ((2, 4, 2, 2, 3, 1, 11), (4, 4, 2, 2, 2, 1, 11), (2, 4, 2, 4, 3, 1, 11), (4, 2, 2, 2, 3, 1, 11), (8, 2, 1, 4, 2, 1, 11)):
    try:
        for i in idx:
            m.set_conv_cfg(i, B, cfg)
    except Exception as e:
        print(cfg, "refused", str(e)[:80]); continue
    t = fwd_ms(m)
    out = m(batch)
    dev_ = max(float((out[k] - ref[k]).abs().max()) for k in ("pred_pose", "pred_shape", "pred_cam", "smpl_vertices"))
    print(f"  14x14 on ALG 11 {cfg}: {t:.3f} ms ({(t / base - 1) * 100:+.2f} %), max deviation from the table forward {dev_:.1e}", flush=True)
for i in idx:
    m.set_conv_cfg(i, B, cur[i])
print(f"back to the table: {fwd_ms(m):.3f} ms")
