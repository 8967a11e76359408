"""Round 6: in-context pass over the 3x3 shapes that run on the split-K direct conv (ALG 5) at small batches: the <4, 4> form of the
table against <1, 8> (cfg.R = 2: 16 pixels per block, 8 K steps in flight) and <2, 6> (cfg.R = 3), timed as hipGraph replays of the
whole forward.   python tools/splitk_tune.py variant B [--write] [--all-s2: also offer the split-K forms to every stride-2 3x3 shape, whatever it runs on]"""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth, tune  # noqa: E402
from poco_amd._lib import PocoHipError  # noqa: E402
from tests import util  # noqa: E402

variant, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine(variant, max_batch=B)
m(batch)


def fwd_ms(reps=80):
    m.release_graphs()
    out = m._alloc_outputs(B, False)
    for _ in range(8):
        m.graph_forward(batch, out)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            m.graph_forward(batch, out)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


shapes = {}
for i, _ in enumerate(m.ops()):
    d = m.conv_desc(i)
    if d is None or d[4] != 3:
        continue
    cfg = tuple(m.conv_cfg(i, B))
    if cfg[6] == 5 or ("--all-s2" in sys.argv and d[5] == 2):
        shapes.setdefault(tune.shape_key(B, *d[:6]), []).append(i)
base = fwd_ms()
print(f"{variant} B={B}: table {base:.4f} ms; {len(shapes)} split-K 3x3 shapes", flush=True)
cur_t, picked = base, {}
for k, idxs in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
    cur = tuple(m.conv_cfg(idxs[0], B))
    best, best_t = cur, cur_t
    for form in ((1, 2, 3) if cur[6] != 5 else (2, 3)):
        for wm in (sorted({cur[2], 16, 8}) if cur[6] == 5 else (4, 8, 16)):
            c = (cur[0], cur[1], wm, cur[3], form, cur[5], 5) if cur[6] == 5 else (1, 1, wm, 1, form, 1, 5)
            try:
                for i in idxs:
                    m.set_conv_cfg(i, B, c)
            except PocoHipError:
                continue
            t = fwd_ms()
            print(f"    {k} x{len(idxs)} {c}: {t:.4f} ms", flush=True)
            if t < best_t * 0.997:
                best, best_t = c, t
    for i in idxs:
        m.set_conv_cfg(i, B, best)
    if best != cur:
        picked[k] = best
        cur_t = best_t
    print(f"  {k:28s} x{len(idxs):3d} {cur} -> {best}  forward {cur_t:.4f} ms", flush=True)
final = fwd_ms()
print(f"{variant} B={B}: {base:.4f} -> {final:.4f} ms ({(final / base - 1) * 100:+.2f} %), {len(picked)} entries moved: {picked}")
if "--write" in sys.argv and picked and final < base * 0.997:
    full = json.loads(tune.TABLE.read_text())
    for k, c in picked.items():
        e = full.setdefault(k, {"uses": len(shapes[k])})
        e["cfg"] = list(c); e["in_context"] = True
    tune.TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
    print("wrote", tune.TABLE)
