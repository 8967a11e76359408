"""The LAST k convs of a branch's chain in every HR module on another configuration (they run alone in front of the module's join):
    python tools/tail_cfg.py variant B branch "cfg" kmax [rounds]"""
import re, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
variant, B, branch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = tuple(int(x) for x in sys.argv[4].split(","))
kmax = int(sys.argv[5]); rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 2
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine(variant, max_batch=B)
m(batch)
mods = {}
for i, (name, _, _) in enumerate(m.ops()):
    mt = re.match(rf"backbone\.(stage\d\.\d+)\.branches\.{branch}\.(\d)\.conv(\d)$", name)
    if mt:
        mods.setdefault(mt.group(1), []).append(i)
table = {i: tuple(m.conv_cfg(i, B)) for v in mods.values() for i in v}
print(f"{len(mods)} modules, {sum(len(v) for v in mods.values())} convs on branch {branch}; table cfg {next(iter(table.values()))}")
out = m._alloc_outputs(B, False)


def fwd_ms(reps=50):
    m.release_graphs()
    for _ in range(6):
        m.graph_forward(batch, out)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            m.graph_forward(batch, out)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


for r in range(rounds):
    for k in range(0, kmax + 1):
        for v in mods.values():
            for j, i in enumerate(v):
                m.set_conv_cfg(i, B, cfg if j >= len(v) - k else table[i])
        print(f"round {r}: last {k} convs of each chain on {cfg}: {fwd_ms():.4f} ms", flush=True)
