#!/usr/bin/env python3
"""Dot interaction forward + backward at B = 65 536 for the MFMA shape (D = 16, n <= 32) and a few
shapes that take the general kernel.  Prints one line per shape: time, algorithmic bytes, TB/s.

    python scripts/bench_interaction.py            # on an MI355X
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.interaction import dot_interaction  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def main():
    _lib.use_native()
    dev = torch.device("cuda", 0)
    B = 65536
    for F, D in ((26, 16), (26, 32), (26, 64), (39, 16), (8, 128), (63, 32)):
        n = F + 1
        dense = torch.randn(B, D, device=dev, requires_grad=True)
        sparse = torch.randn(B, F * D, device=dev, requires_grad=True)
        out = dot_interaction(dense, sparse, D, True, True)
        go = torch.randn_like(out)
        t_f = timed(lambda: dot_interaction(dense, sparse, D, True, True))

        def fb():
            o = dot_interaction(dense, sparse, D, True, True)
            o.backward(go)
            dense.grad = sparse.grad = None

        t_fb = timed(fb)
        width = out.shape[1]
        by_f = 4 * B * (n * D + width)
        by_b = 4 * B * (n * D + width + n * D)
        print(f"n={n:3d} D={D:4d} {'mfma' if D == 16 and n <= 32 else 'general':8s} fwd {t_f:7.1f} us = {by_f / t_f / 1e6:5.2f} TB/s   "
              f"bwd {t_fb - t_f:7.1f} us = {by_b / max(t_fb - t_f, 1e-3) / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
