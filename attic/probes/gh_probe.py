import ctypes as C, os, sys
sys.path.insert(0, '/root/repo')
from poco_amd._lib import check, lib
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
cands = [(4, 4, 2, 2, 8, 1, 12)]
flat = (C.c_int * 7)(*cands[0]); ms = (C.c_float * 1)()
for H, Cin, Cout in [(7, 512, 2048), (28, 512, 128), (14, 256, 1024)]:
    check(L.poco_tune_conv(64, H, H, Cin, Cout, 1, 1, flat, 1, 20, ms, None), "tune")
    print(os.environ.get("POCO_GH_DBG", "0"), H, Cin, Cout, "%.1f us" % (ms[0] * 1e3))
