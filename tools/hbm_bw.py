import torch, time
x = torch.randn(64*56*56*256, device='cuda')   # 205 MB
y = torch.empty_like(x)
z = torch.empty(64*56*56*64, device='cuda')
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
gb = x.numel()*4/1e9
print("copy 205MB->205MB: %.1f us = %.2f TB/s (r+w)" % (t(lambda: y.copy_(x))*1e6, 2*gb/t(lambda: y.copy_(x))/1e3))
print("sum  205MB read : %.1f us = %.2f TB/s" % (t(lambda: x.sum())*1e6, gb/t(lambda: x.sum())/1e3))
print("relu_ in place   : %.1f us = %.2f TB/s (r+w)" % (t(lambda: y.relu_())*1e6, 2*gb/t(lambda: y.relu_())/1e3))
print("add 3 tensors    : %.1f us = %.2f TB/s" % (t(lambda: torch.add(x, y, out=y))*1e6, 3*gb/t(lambda: torch.add(x, y, out=y))/1e3))
