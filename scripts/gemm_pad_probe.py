#!/usr/bin/env python3
"""Does padding K = 783 -> 784 (16-byte aligned rows) help the three GEMMs of the 783->64 layer?"""
import os
import sys

import torch
import torch.cuda.tunable as tn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
dev = torch.device("cuda", 0)
tn.enable(True)
tn.tuning_enable(True)
tn.set_filename(f"/tmp/pad_probe_{os.getpid()}.csv", insert_device_ordinal=False)
B = 65536


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K in (783, 784, 800):
    x = torch.randn(B, K, device=dev)
    w = torch.randn(64, K, device=dev)
    b = torch.randn(64, device=dev)
    g = torch.randn(B, 64, device=dev)
    fwd = t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))
    fwd2 = t(lambda: torch.nn.functional.linear(x, w, b))
    dgrad = t(lambda: g @ w)
    wgrad = t(lambda: g.t() @ x)
    # same products on a [B, 784]-strided view holding K valid columns
    print(f"K={K}: fwd(relu epilogue)={fwd:.1f} fwd(linear)={fwd2:.1f} dgrad={dgrad:.1f} wgrad={wgrad:.1f} us", flush=True)
xs = torch.randn(B, 784, device=dev)[:, :783]
w = torch.randn(64, 783, device=dev)
g = torch.randn(B, 64, device=dev)
print(f"K=783 in 784-strided rows: fwd={t(lambda: torch.nn.functional.linear(xs, w)):.1f} wgrad={t(lambda: g.t() @ xs):.1f} us")
