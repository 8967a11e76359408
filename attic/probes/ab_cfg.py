"""A/B a tile configuration for one conv shape inside the whole forward (4 lanes, graph replay):
python tools/ab_cfg.py HxWxCinxCoutkKsS "cfgA" "cfgB" [variant] [B]   (cfg = 7 comma-separated ints)"""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import synth, tune  # noqa: E402
from tests import util  # noqa: E402

shape = sys.argv[1]
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:4]]
variant = sys.argv[4] if len(sys.argv) > 4 else "hrnet_w48_cls-cliff"
B = int(sys.argv[5]) if len(sys.argv) > 5 else 64
batch = util.cuda_batch(synth.synth_batch(B, 1), torch.device("cuda:0"))
for rep in range(2):
    for cfg in cfgs:
        m = util.make_engine(variant, max_batch=B)
        m(batch)
        n = 0
        for i, _ in enumerate(m.ops()):
            d = m.conv_desc(i)
            if d is not None and tune.shape_key(B, *d[:6]).split("x", 1)[1] == shape:
                m.set_conv_cfg(i, B, cfg); n += 1
        out = m._alloc_outputs(B, False)
        for _ in range(5):
            m.graph_forward(batch, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            m.graph_forward(batch, out)
        torch.cuda.synchronize()
        print(f"{shape} cfg={cfg} ({n} ops): {(time.perf_counter()-t0)/30*1e3:.3f} ms/forward")
        del m
