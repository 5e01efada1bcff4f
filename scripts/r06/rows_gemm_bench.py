#!/usr/bin/env python3
"""The tall-input Linear kernels (csrc/gemm_rows.hip) against the GEMM library on the DIN attention MLP's shapes at Taobao scale
(N = 458 752 positions): forward layers, input gradient, weight gradients; microseconds per launch (HIP events, median of 30)
and the fraction of the exact-fp32 MFMA peak (157.3 TFLOP/s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.dense import linear_rows, linear_rows_wgrad, weight_grad  # noqa: E402

if os.environ.get("ROWS_LIB"):  # a variant build (scripts/r06/build_variant.py) instead of the product library
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), os.environ["ROWS_LIB"]))
    print("library:", os.environ["ROWS_LIB"], flush=True)
CHECK = not os.environ.get("ROWS_NO_CHECK")
dev = torch.device("cuda", 0)
N = int(os.environ.get("ROWS_N", 458752))
PEAK = 157.3e12


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def line(name, flop, own, lib):
    print("%-44s own %7.1f us (%.3f of peak)   library %7.1f us (%.3f)   x%.2f" % (name, own, flop / own / 1e-6 / PEAK, lib, flop / lib / 1e-6 / PEAK,
                                                                                  lib / own), flush=True)


torch.manual_seed(0)
for K, H in ((144, 256), (96, 256), (256, 64), (192, 256), (64, 256), (128, 64)):
    x = torch.randn(N, K, device=dev)
    W = torch.randn(H, K, device=dev) / K ** 0.5
    b = torch.randn(H, device=dev)
    own = timeit(lambda: linear_rows(x, W, b, relu=True))
    lib = timeit(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=False))
    line("forward  [%d, %d] x [%d, %d]^T + relu" % (N, K, H, K), 2.0 * N * K * H, own, lib)
    err = (linear_rows(x[:4096], W, b, relu=True) - torch.relu(x[:4096] @ W.t() + b)).abs().max().item()
    assert err < 1e-3 or not CHECK, err
seg = (torch.arange(N, device=dev) // 56).to(torch.int32)
rv = torch.randn(int(seg.max()) + 1, 256, device=dev)
x = torch.randn(N, 144, device=dev)
W = torch.randn(256, 96, device=dev) / 10
own = timeit(lambda: linear_rows(x, W, None, relu=True, rowvec=rv, row_index=seg, K=96))
print("forward  K = 96 of 144-wide rows + per-sample row vector + relu: %7.1f us" % own, flush=True)
for K, H in ((256, 144), (256, 96), (64, 256), (256, 48)):
    g = torch.randn(N, K, device=dev)
    W = torch.randn(K, H, device=dev) / K ** 0.5
    own = timeit(lambda: linear_rows(g, W, out_major=False))
    lib = timeit(lambda: g @ W)
    line("input gradient [%d, %d] x [%d, %d]" % (N, K, K, H), 2.0 * N * K * H, own, lib)
for H, K in ((256, 144), (256, 96), (64, 256), (256, 48), (128, 128)):
    g = torch.randn(N, H, device=dev)
    x = torch.randn(N, K, device=dev)
    own = timeit(lambda: linear_rows_wgrad(g, x))
    lib = timeit(lambda: weight_grad(g, x))
    lib2 = timeit(lambda: g.t() @ x)
    line("weight gradient [%d, %d]^T x [%d, %d]" % (N, H, N, K), 2.0 * N * K * H, own, min(lib, lib2))
