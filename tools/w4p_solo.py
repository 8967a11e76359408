"""Solo timings of the W48 F(4x4) shapes (rectangular / flat items) next to two control convs that do not run on ALG 8, for
same-box A/B runs of alternative builds:  POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_<x>.so python tools/w4p_solo.py [B] [fwd]"""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import ops, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
rows = []
for (H, W, Cin, Cout, ks, st), cfgs in [
        ((56, 56, 48, 48, 3, 1), [(1, 3, 2, 4, 8, 1, 8), (1, 3, 2, 4, 4, 0, 8)]),
        ((28, 28, 96, 96, 3, 1), [(1, 3, 2, 4, 16, 1, 8), (1, 3, 2, 4, 4, 0, 8)]),
        ((14, 14, 192, 192, 3, 1), [(2, 3, 2, 4, 16, 2, 8), (1, 3, 2, 4, 16, 2, 8)]),
        ((56, 56, 64, 64, 3, 1), [(1, 2, 2, 4, 8, 1, 8), (1, 2, 2, 4, 4, 0, 8)]),
        ((28, 28, 96, 192, 3, 2), [None]),            # control: stride-2 gather GEMM / LDS-staged conv (heuristic)
        ((14, 14, 1024, 512, 1, 1), [None])]:         # control: 1x1 GEMM
    x = torch.randn(B, H, W, Cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, ks, ks)) / np.sqrt(ks * ks * Cin)).astype(np.float32)
    for cfg in cfgs:
        try:
            ts = [ops.bench_conv2d(x, w, st, cfg=cfg, iters=40)[0] * 1e3 for _ in range(3)]
            rows.append(f"{H}x{W} {Cin}->{Cout} k{ks}s{st} {cfg}: {min(ts):.1f} us")
        except Exception as e:
            rows.append(f"{H}x{W} {Cin}->{Cout} k{ks}s{st} {cfg}: refused")
print("\n".join(rows), flush=True)
if len(sys.argv) > 2:
    from tests import util
    for variant, Bv in (("hrnet_w48_cls-cliff", B), ("hrnet_w32-pare", 32)):
        m = util.make_engine(variant, max_batch=Bv)
        batch = util.cuda_batch(synth.synth_batch(Bv, 1), dev)
        out = m._alloc_outputs(Bv, False)
        for _ in range(8):
            m.graph_forward(batch, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            m.graph_forward(batch, out)
        torch.cuda.synchronize()
        print(f"{variant} B={Bv} forward (table): {(time.perf_counter() - t0) / 40 * 1e3:.3f} ms")
