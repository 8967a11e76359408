"""Round 4: ALG 11 only on the LAST convs of every 14x14 branch chain (the convs that run alone in front of a module's join), ALG 8 on
the others.  python tools/wg14_tail.py [variant] [B]"""
import re, sys, time
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
m(batch)
chains = {}            # (stage, module, branch) -> [(position in the chain 0..7, op index)]
for i, (nm, _, _) in enumerate(m.ops()):
    d = m.conv_desc(i)
    mm = re.search(r"stage(\d)\.(\d)\.branches\.(\d)\.(\d)\.conv(\d)$", nm)
    if d is None or not mm or d[0] != 14 or d[4] != 3:
        continue
    st, mod, br, blk, cv = map(int, mm.groups())
    chains.setdefault((st, mod, br), []).append((2 * blk + cv - 1, i))
cur = {i: m.conv_cfg(i, B) for ch in chains.values() for _, i in ch}
base = fwd_ms(m)
print(f"{variant} B={B}: {len(chains)} 14x14 chains, table {base:.3f} ms", flush=True)
for cfg in ((4, 2, 2, 2, 3, 1, 11), (2, 4, 2, 4, 3, 1, 11)):
    for k0 in (7, 6, 5, 4, 2):
        for ch in chains.values():
            for pos, i in ch:
                m.set_conv_cfg(i, B, cfg if pos >= k0 else cur[i])
        t = fwd_ms(m)
        print(f"  ALG 11 {cfg} on chain positions >= {k0}: {t:.3f} ms ({(t / base - 1) * 100:+.2f} %)", flush=True)
