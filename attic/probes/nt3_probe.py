import ctypes as C, sys
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from poco_amd._lib import check, lib
from poco_amd import ops
from tests.test_conv_gpu import _conv_fp64_gpu
torch.cuda.set_device(0)
L=lib()
L.poco_tune_conv.argtypes=[C.c_int]*7+[C.POINTER(C.c_int),C.c_int,C.c_int,C.POINTER(C.c_float),C.c_void_p]
dev=torch.device('cuda:0')
for (B,H,W,Cin,Cout,R,NI) in [(32,56,56,128,128,8,1),(32,56,56,480,128,8,1),(32,56,56,256,256,8,1),(64,28,28,128,128,16,1),(64,14,14,256,256,16,2),(32,28,28,256,256,16,1),(32,28,28,128,128,16,1)]:
    res=[]
    for NT in (2,3):
        cfg=(1,NT,2,4,R,NI,8)
        flat=(C.c_int*7)(*cfg); ms=(C.c_float*1)()
        check(L.poco_tune_conv(B,H,W,Cin,Cout,3,1,flat,1,20,ms,None),"tune")
        res.append(ms[0]*1e3)
    # parity at NT=3 (ragged last n-group)
    x=torch.randn((B,H,W,Cin),device=dev)
    w=(np.random.default_rng(0).standard_normal((Cout,Cin,3,3))/np.sqrt(Cin*9)).astype(np.float32)
    r=torch.randn((B,H,W,Cout),device=dev)
    out=ops.conv2d_nhwc(x,w,None,np.zeros(Cout,np.float32),1,r,True,cfg=(1,3,2,4,R,NI,8))
    ref=_conv_fp64_gpu(x,w,np.zeros(Cout,np.float32),1,r,True)
    dev_=float((out.double()-ref).abs().max())/max(1.0,float(ref.abs().max()))
    print(f"{B}x{H}x{W} {Cin}->{Cout}: NT=2 {res[0]:.1f} us, NT=3 {res[1]:.1f} us, NT=3 max rel dev {dev_:.1e}")
