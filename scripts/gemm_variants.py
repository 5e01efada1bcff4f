#!/usr/bin/env python3
"""The three 783 <-> 64 products of DLRM's top MLP at B = 65536 (fp32), in several formulations, timed under TunableOp."""
import os, sys, time
import torch
import torch.cuda.tunable as tn
tn.enable(True); tn.tuning_enable(True)
tn.set_filename("/tmp/gv_tunable.csv", insert_device_ordinal=False)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
z = torch.randn(B, 783, device=dev); g1 = torch.randn(B, 64, device=dev); W1 = torch.randn(64, 783, device=dev); b1 = torch.randn(64, device=dev)
z784 = torch.zeros(B, 784, device=dev); z784[:, :783] = z; zv = z784[:, :783]
W1p = torch.zeros(64, 784, device=dev); W1p[:, :783] = W1
def t(name, fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:58s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us", flush=True)
t("fwd  addmm_activation(b1, z, W1.t())", lambda: torch._addmm_activation(b1, z, W1.t(), use_gelu=False))
t("fwd  addmm_activation(b1, z[stride 784], W1.t())", lambda: torch._addmm_activation(b1, zv, W1.t(), use_gelu=False))
t("fwd  z784 @ W1p.t() (K = 784)", lambda: z784 @ W1p.t())
t("dZ   g1 @ W1", lambda: g1 @ W1)
t("dZ   g1 @ W1p (N = 784)", lambda: g1 @ W1p)
t("dZ   (W1.t() @ g1.t()).t()", lambda: (W1.t() @ g1.t()).t())
t("dW   g1.t() @ z", lambda: g1.t() @ z)
t("dW   (z.t() @ g1).t()", lambda: (z.t() @ g1).t())
t("dW   g1.t() @ z784 (N = 784)", lambda: g1.t() @ z784)
for S in (8, 16, 32, 64):
    t(f"dW   bmm split-K {S} + sum", lambda S=S: torch.bmm(g1.view(S, B // S, 64).transpose(1, 2), z.view(S, B // S, 783)).sum(0))
    t(f"dW   bmm split-K {S} (N = 784) + sum", lambda S=S: torch.bmm(g1.view(S, B // S, 64).transpose(1, 2), z784.view(S, B // S, 784)).sum(0))
for S in (8, 16, 32, 64, 128):
    W1t = W1.t().contiguous()
    t(f"fwd  bmm {S} slices of the batch (no epilogue)", lambda S=S: torch.bmm(z.view(S, B // S, 783), W1t.expand(S, 783, 64)))
    t(f"dZ   bmm {S} slices of the batch", lambda S=S: torch.bmm(g1.view(S, B // S, 64), W1.expand(S, 64, 783)))
t("fwd  F.linear(z, W1, b1)", lambda: torch.nn.functional.linear(z, W1, b1))
