import os, sys, torch
sys.path.insert(0, '/root/repo')
from torcheasyrec_amd.dense import linear_rows, linear_rows_wgrad, weight_grad
dev = torch.device("cuda", 0)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)*1e3)
    ts.sort(); return ts[len(ts)//2]
for N in (8192, 16384, 65536):
    for K,H in ((128,256),(256,128),(128,64),(256,256),(64,128)):
        x=torch.randn(N,K,device=dev); W=torch.randn(H,K,device=dev)/K**.5; b=torch.randn(H,device=dev)
        own=timeit(lambda: linear_rows(x,W,b,relu=True)); lib=timeit(lambda: torch._addmm_activation(b,x,W.t(),use_gelu=False))
        g=torch.randn(N,H,device=dev)
        ownd=timeit(lambda: linear_rows(g,W,out_major=False)); libd=timeit(lambda: g@W)
        oww=timeit(lambda: linear_rows_wgrad(g,x)); libw=timeit(lambda: g.t()@x)
        print(f"N {N:6d} {K:3d}->{H:3d}: fwd own {own:6.1f} lib {lib:6.1f} | dX own {ownd:6.1f} lib {libd:6.1f} | dW own {oww:6.1f} lib {libw:6.1f}", flush=True)
