import os, sys, torch
sys.path.insert(0, os.getcwd())
from autodiffusion_amd import ops
DEV="cuda:0"
def timeit(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
n,t,h,d=256,1024,6,64
qkv = torch.randn(n, t, 3*h*d, device=DEV).to(torch.bfloat16)
ms = timeit(lambda: ops.attention(qkv, h, True))
print(f"{os.environ.get('ADM_HIP_LIB','default').split('/')[-1]:20s} T=1024 H=6 fwd {ms*1e3:8.1f} us")
