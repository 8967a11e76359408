"""Call the stand-alone SMPL LBS operator a few times (for rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests import util  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = util.make_engine("resnet50-cliff", max_batch=B)
betas = torch.randn(B, 10, device="cuda")
rot = torch.linalg.qr(torch.randn(B, 24, 3, 3, device="cuda"))[0].contiguous()
for _ in range(20):
    m.smpl_lbs(betas, rot)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    m.smpl_lbs(betas, rot)
e1.record(); torch.cuda.synchronize()
print("smpl_lbs us/call", e0.elapsed_time(e1) / 50 * 1e3)
