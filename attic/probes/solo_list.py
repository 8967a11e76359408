"""Solo times of every legal configuration of one conv shape, fastest first: python tools/solo_list.py B H W Cin Cout ks stride [top]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd import tune  # noqa: E402
from poco_amd._lib import lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
B, H, W, Cin, Cout, ks, st = map(int, sys.argv[1:8])
top = int(sys.argv[8]) if len(sys.argv) > 8 else 15
res = sorted(r for r in tune.solo_times(L, B, H, W, Cin, Cout, ks, st, iters=20) if r[0] > 0)
for t, c in res[:top]:
    print(f"{t*1e3:8.1f} us  {tuple(c)}")
print(len(res), "valid configurations")
