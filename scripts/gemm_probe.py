#!/usr/bin/env python3
"""Time the DLRM-Criteo dense part (MLP fwd+bwd, fp32, B=65536) under the current BLAS settings."""
import os, sys, time, torch
torch.manual_seed(0)
dev = torch.device("cuda")
B = int(os.environ.get("B", 65536))
def mlp(i, hs):
    L = []
    for h in hs:
        L += [torch.nn.Linear(i, h), torch.nn.ReLU()]; i = h
    return torch.nn.Sequential(*L).to(dev)
bot, top, out = mlp(13, [64, 16]), mlp(783, [64, 32]), torch.nn.Linear(32, 1).to(dev)
x = torch.randn(B, 13, device=dev); z = torch.randn(B, 783, device=dev, requires_grad=True)
def step():
    d = bot(x); y = out(top(z)).squeeze(1)
    (y.sum() + d.sum()).backward()
for _ in range(int(os.environ.get("WARM", 10))): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[1] if len(sys.argv)>1 else 'default'}: {e0.elapsed_time(e1)/20*1e3:.0f} us per dense fwd+bwd (B={B})")
