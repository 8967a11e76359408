import sys
sys.path.insert(0, '.')
import numpy as np, torch
from poco_amd import ops
dev = torch.device("cuda:0")
for (B, H, W, Cin, Cout), cfgs in [((32, 56, 56, 480, 128), [(1, 3, 2, 4, 4, 0, 8), (2, 3, 2, 4, 4, 0, 8)]), ((64, 14, 14, 192, 192), [(2, 3, 2, 4, 16, 2, 8)]),
                                   ((64, 28, 28, 96, 96), [(1, 3, 2, 4, 4, 0, 8), (1, 3, 2, 4, 16, 1, 8)]), ((32, 56, 56, 128, 128), [(1, 3, 2, 4, 8, 1, 8)]),
                                   ((64, 56, 56, 48, 48), [(1, 3, 2, 4, 4, 0, 8)])]:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    for cfg in cfgs:
        ts = [ops.bench_conv2d(x, w, 1, cfg=cfg, iters=40)[0] * 1e3 for _ in range(3)]
        print(f"{B}x{H}x{W} {Cin}->{Cout} {cfg}: {min(ts):.1f} us", flush=True)
