"""Phase times (shader clocks, s_memtime) of one MFMA wave and one producer wave of block 0 of the ALG 8 kernel.
Needs the -DW4P_TRACE=1 build:  bash tools/build_exp.sh conv_wino4p.hip W4P_TRACE 1 ; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_W4P_TRACE_1.so python tools/w4p_trace.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
L.poco_w4p_trace.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for (B, H, W, Cin, Cout), cfg in [((64, 56, 56, 48, 48), (1, 3, 2, 4, 8, 1, 8)), ((64, 56, 56, 48, 48), (1, 3, 2, 4, 4, 0, 8)), ((64, 14, 14, 192, 192), (2, 3, 2, 4, 16, 2, 8)), ((64, 28, 28, 96, 96), (1, 3, 2, 4, 16, 1, 8))]:
    flat = (C.c_int * 7)(*cfg); ms = (C.c_float * 1)()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, 3, 1, flat, 1, 5, ms, None), "tune")
    buf = (C.c_ulonglong * 96)()
    assert L.poco_w4p_trace(buf, 96) == 0
    S = int(buf[4]) or 1
    print(f"{H}x{W} {Cin}->{Cout}: {ms[0]*1e3:.1f} us/launch; S = {S} slices per item")
    print(f"  MFMA wave 0: prologue wait {buf[0]} clk | per slice: work {buf[1]/S:.0f} clk, barrier wait {buf[2]/S:.0f} clk | epilogue {buf[3]} clk")
    names = ["transform burst", "store U (+wait loads)", "load U + raw DMA issue", "window reads (issue)", "wait_vm", "barrier wait"]
    print("  producer 0 per slice: " + " | ".join(f"{n} {buf[8+k]/S:.0f}" for k, n in enumerate(names)))
    print("  all MFMA waves (work / barrier wait per slice): " + "  ".join(f"w{w}: {buf[16+2*w]/S:.0f}({buf[48+w]/S:.0f} in wait_vm)/{buf[17+2*w]/S:.0f}" for w in range(8)))
    for w, o in ((0, 56), (5, 64)):
        print(f"  exchange rounds of MFMA wave {w} (sum over the rounds): barrier A {buf[o]} | Z + writes {buf[o+1]} | barrier B {buf[o+2]} | reads + Y (round 4: whole rest) {buf[o+3]} | next loads awaited {buf[o+4]} | stores issued {buf[o+5]}")
    print("  all producers (transform+V store / window reads / barrier wait per slice): " + "  ".join(f"p{w}: {buf[32+4*w]/S:.0f}/{buf[33+4*w]/S:.0f}/{buf[34+4*w]/S:.0f}" for w in range(4)))
