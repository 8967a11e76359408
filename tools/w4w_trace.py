"""Phase times (s_memtime) of the waves of block 0 of the ALG 13 kernel over its first item.  Needs a -DW4W_TRACE=1 build:
bash tools/build_exp.sh conv_wino4w.hip W4W_TRACE 1; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_W4W_TRACE_1.so python tools/w4w_trace.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
L.poco_w4w_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for (B, H, W, Cin, Cout), cfg in [((64, 56, 56, 48, 48), (1, 3, 2, 1, 8, 1, 13)), ((64, 14, 14, 192, 192), (2, 3, 2, 1, 16, 2, 13)), ((64, 56, 56, 64, 64), (1, 2, 2, 1, 8, 1, 13))]:
    flat = (C.c_int * 7)(*cfg); ms = (C.c_float * 1)()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, 3, 1, flat, 1, 5, ms, None), "tune")
    buf = (C.c_ulonglong * 64)()
    assert L.poco_w4w_trace(buf, 64) == 0
    S = int(buf[63]) or 1
    nt = cfg[1]
    print(f"{H}x{W} {Cin}->{Cout} NT={nt}: {ms[0]*1e3:.1f} us/launch; S = {S}")
    for w in range(2 * nt):
        print(f"  MFMA wave {w}: per slice issue+MFMA {buf[4*w]/S:.0f} | wait_vm {buf[4*w+1]/S:.0f} | barrier {buf[4*w+2]/S:.0f} | epilogue {buf[4*w+3]} = set-up {buf[48+2*w]} + first stage {buf[49+2*w]} + columns / stores {buf[4*w+3]-buf[48+2*w]-buf[49+2*w]}")
    for pw in range(2):
        b = 32 + 5 * pw
        print(f"  producer {pw}: per slice LDS-DMA requests {buf[b]/S:.0f} | transform + V store {buf[b+1]/S:.0f} | window reads {buf[b+2]/S:.0f} | wait_vm {buf[b+3]/S:.0f} | barrier {buf[b+4]/S:.0f}")
    if nt == 2:
        for dw in range(2):
            print(f"  patch requester {dw}: per slice requests {buf[58+2*dw]/S:.0f} | wait + barrier {buf[59+2*dw]/S:.0f}")
