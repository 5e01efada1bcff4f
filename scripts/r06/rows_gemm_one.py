#!/usr/bin/env python3
"""One of the tall-input Linear kernels alone, a few launches (the workload of a rocprofv3 --pmc pass):
   rows_gemm_one.py fwd|rv|dx|wg  K H [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd.dense import linear_rows, linear_rows_wgrad  # noqa: E402

kind, K, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 458752
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if kind in ("fwd", "rv"):
    x = torch.randn(N, K, device=dev)
    W = torch.randn(H, K, device=dev) / K ** 0.5
    b = torch.randn(H, device=dev)
    seg = (torch.arange(N, device=dev) // 56).to(torch.int32)
    rv = torch.randn(int(seg.max()) + 1, H, device=dev)
    fn = (lambda: linear_rows(x, W, b, relu=True)) if kind == "fwd" else (lambda: linear_rows(x, W, None, relu=True, rowvec=rv, row_index=seg))
elif kind == "dx":
    g = torch.randn(N, K, device=dev)
    W = torch.randn(K, H, device=dev) / K ** 0.5
    fn = lambda: linear_rows(g, W, out_major=False)
else:
    g = torch.randn(N, H, device=dev)
    x = torch.randn(N, K, device=dev)
    fn = lambda: linear_rows_wgrad(g, x)
for _ in range(5):
    fn()
torch.cuda.synchronize()
