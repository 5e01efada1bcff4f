#!/usr/bin/env python3
"""tzr_linear_bwd_relu against the three ops it replaces (GEMM, ReLU mask, column sums = torch matmul + tzr_relu_bwd_colsum) at
the shape of DIN's attention MLP on the Taobao config: N = 450 560 positions, K = 64, H = 256; and tzr_head_bwd_relu against
tzr_head_bwd + tzr_relu_bwd_colsum at [N, 64]."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.dense import head_bwd, head_bwd_relu, linear_bwd_relu, relu_bwd_colsum  # noqa: E402

_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 450560


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K, H in ((64, 256), (64, 128), (32, 64)):
    g = torch.randn(N, K, device=dev)
    W = torch.randn(K, H, device=dev) / 8
    y = torch.relu(torch.randn(N, H, device=dev))
    t_new = timed(lambda: linear_bwd_relu(g, W, y))
    t_old = timed(lambda: relu_bwd_colsum(g @ W, y))
    a, ca = linear_bwd_relu(g, W, y)
    b, cb = relu_bwd_colsum(g @ W, y)
    err = float((a - b).abs().max()) / float(b.abs().max())
    nbytes = 4.0 * N * (K + 2 * H)
    print(f"N {N} K {K} H {H}: tzr_linear_bwd_relu {t_new:.1f} us = {nbytes / t_new / 1e6:.2f} TB/s of g_in + y + g_out, "
          f"{2.0 * N * K * H / t_new / 1e6:.1f} TFLOP/s; matmul + tzr_relu_bwd_colsum {t_old:.1f} us; max |diff| / max {err:.2e}; "
          f"colsum diff {float((ca - cb).abs().max()):.2e}", flush=True)
x = torch.relu(torch.randn(N, 64, device=dev))
w = torch.randn(1, 64, device=dev)
gy = torch.randn(N, device=dev) / N
t_new = timed(lambda: head_bwd_relu(gy, x, w))


def old():
    dh, _, _ = head_bwd(gy, x, w, True)
    return relu_bwd_colsum(dh, x)


print(f"[N, 64]: tzr_head_bwd_relu {t_new:.1f} us; tzr_head_bwd + tzr_relu_bwd_colsum {timed(old):.1f} us", flush=True)
