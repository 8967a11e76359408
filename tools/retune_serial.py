"""Solo sweep (unbiased slots, tune._timed) of every conv shape of a variant against the table's entry, the best few candidates then tried
inside the hipGraph forward; winners go to gpurun_out/retune_<variant>_<B>.json.  For chains of kernels (ResNet-50) and for the serial
sections of the HRNet variants a solo gain is a forward gain.   python tools/retune_serial.py variant B [min_us]"""
import ctypes as C
import json
import sys
import time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth, tune  # noqa: E402
from poco_amd._lib import lib  # noqa: E402
from tests import util  # noqa: E402

variant, B = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
dev = torch.device("cuda:0")
batch = util.cuda_batch(synth.synth_batch(B, 1), dev)
m = util.make_engine(variant, max_batch=B)
m(batch)
shapes = {}
for i, _ in enumerate(m.ops()):
    d = m.conv_desc(i)
    if d is not None and d[0] * d[1] > 1:
        shapes.setdefault(tuple(d[:6]), []).append(i)
out = m._alloc_outputs(B, False)


def fwd_ms(reps=40, rounds=3):
    m.release_graphs()
    for _ in range(6):
        m.graph_forward(batch, out)
    best = 1e9
    for _ in range(rounds):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            m.graph_forward(batch, out)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


cur = fwd_ms()
print(f"{variant} B={B}: {len(shapes)} conv shapes, forward {cur:.4f} ms", flush=True)
updates = {}
for (H, W, Cin, Cout, ks, st), idxs in sorted(shapes.items(), key=lambda kv: -len(kv[1])):
    tcfg = tuple(m.conv_cfg(idxs[0], B))
    cands = [c for c in tune.candidates(B, H, W, Cin, Cout, ks, st) if tuple(c) != tcfg]
    ms = tune._timed(L, B, H, W, Cin, Cout, ks, st, [tcfg] + cands, 8)
    if ms[0] * 1e3 * len(idxs) < min_us:
        continue
    top = sorted((i for i in range(1, len(ms)) if ms[i] > 0), key=lambda i: ms[i])[:5]
    # ALG 6 load schedules for its best tilings
    g6 = [cands[i - 1] for i in top if cands[i - 1][6] == 6 and cands[i - 1][5] == 1][:3]
    extra = [c[:5] + (ni, 6) for c in g6 for ni in range(2, 7) if tune.G1_SCHED_G[ni] * (c[0] + c[1]) <= 4 * c[0] * c[1]]
    pool = [tcfg] + [cands[i - 1] for i in top] + extra
    ms2 = tune._timed(L, B, H, W, Cin, Cout, ks, st, pool, 30)
    order = sorted(range(1, len(pool)), key=lambda i: ms2[i] if ms2[i] > 0 else 1e9)[:2]
    line = f"  {H}x{W} {Cin}->{Cout} k{ks}s{st} x{len(idxs)}: table {tcfg} {ms2[0] * 1e3:.1f} us | best {pool[order[0]]} {ms2[order[0]] * 1e3:.1f} us ({100 * (ms2[0] / ms2[order[0]] - 1):+.1f} %)"
    if ms2[order[0]] > 0.98 * ms2[0]:
        print(line + " kept", flush=True)
        continue
    best_cfg, best_t = None, cur
    for i in order:
        if ms2[i] > 0.985 * ms2[0]:
            continue
        try:
            for j in idxs:
                m.set_conv_cfg(j, B, pool[i])
        except Exception as e:        # a candidate the engine's op does not accept (channel slices)
            continue
        t = fwd_ms()
        if t < best_t - 0.0005 * cur:
            best_cfg, best_t, best_i = pool[i], t, i
    for j in idxs:
        m.set_conv_cfg(j, B, best_cfg if best_cfg else tcfg)
    print(line + f" | forward {cur:.4f} -> {best_t:.4f} ms {'TAKEN ' + str(best_cfg) if best_cfg else 'kept'}", flush=True)
    if best_cfg:
        cur = best_t
        fl = 2.0 * B * ((H - 1) // st + 1) * ((W - 1) // st + 1) * Cin * Cout * ks * ks
        updates[tune.shape_key(B, H, W, Cin, Cout, ks, st)] = {"cfg": list(best_cfg), "ms": round(float(ms2[best_i]), 5), "tflops": round(fl / ms2[best_i] / 1e9, 1),
                                                                "in_context": True, "uses": len(idxs)}
print(f"final forward {fwd_ms():.4f} ms; {len(updates)} entries")
Path("gpurun_out").mkdir(exist_ok=True)
Path(f"gpurun_out/retune_{variant}_{B}.json").write_text(json.dumps(updates, indent=0, sort_keys=True))
print(json.dumps(updates))
