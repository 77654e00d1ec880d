import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from autodiffusion_amd import ops
DEV="cuda:0"
for name,n,hw,cin,cout,pro,res in [("clf proj 256->256 @32",256,32,256,256,0,True),("clf qkv 256->768 @32",256,32,256,768,1,False),("unet proj 384->384 @32",256,32,384,384,0,True),("unet qkv 384->1152 @32",256,32,384,1152,1,False)]:
    x=torch.randn(n,hw,hw,cin,device=DEV).to(torch.bfloat16)
    w=torch.randn(cout,cin,1,1,device=DEV)*cin**-0.5
    wp=ops.pack_conv_weight(w); b=torch.randn(cout,device=DEV)*0.1
    aff=(1+0.1*torch.randn(n,cin,device=DEV),0.1*torch.randn(n,cin,device=DEV)) if pro else None
    r=torch.randn(n,hw,hw,cout,device=DEV).to(torch.bfloat16) if res else None
    out=torch.empty(n,hw,hw,cout,dtype=torch.bfloat16,device=DEV)
    f=lambda: ops.conv(x,wp,b,cout,1,aff=aff,silu=False,res=r,out=out)
    t0=time.perf_counter()
    while time.perf_counter()-t0<0.03: f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*50
    byt=(x.numel()+out.numel()*(2 if res else 1))*2
    print(f"{name:26s} {us:8.1f} us {2.0*n*hw*hw*cin*cout/us/1e6:7.1f} TFLOP/s {byt/us/1e6:6.2f} TB/s")
