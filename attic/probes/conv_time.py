"""Time one conv shape with one configuration: tools/conv_time.py B H W Cin Cout ks stride MT NT WM WN R NI ALG
(combine with POCO_CONV_REPEAT / POCO_CONV_DBG to split K-loop time from the fixed cost)."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
a = [int(x) for x in sys.argv[1:15]]
flat = (C.c_int * 7)(*a[7:14])
ms = (C.c_float * 1)()
check(L.poco_tune_conv(*a[:7], flat, 1, 20, ms, None), "tune")
print(f"{ms[0]*1e3:.1f}")
