"""Same-box A/B of one shape's configuration inside the whole forward (hipGraph replays, alternating, `rounds` times):
  python tools/ab_shape_cfg.py variant B HxWxCinxCout "cfgA" "cfgB" [rounds] [engine options "k=v,k=v"]"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
variant, B = sys.argv[1], int(sys.argv[2])
H, W, Cin, Cout = map(int, sys.argv[3].split("x"))
cfgs = [tuple(int(x) for x in c.split(",")) for c in sys.argv[4:6]]
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine(variant, max_batch=B, options=sys.argv[7] if len(sys.argv) > 7 else None)
m(batch)
idxs = [i for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None and tuple(m.conv_desc(i)[:6]) == (H, W, Cin, Cout, 3, 1)]
print(f"{len(idxs)} ops of shape {H}x{W} {Cin}->{Cout}; table cfg {tuple(m.conv_cfg(idxs[0], B))}")


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


for r in range(rounds):
    for c in cfgs:
        for i in idxs:
            m.set_conv_cfg(i, B, c)
        print(f"round {r} {c}: {fwd_ms():.3f} ms", flush=True)
