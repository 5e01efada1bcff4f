#!/usr/bin/env python3
"""The D = 16 dot-interaction backward at the DLRM-Criteo shape (n = 27, B = 65 536) under its tzr_tune knobs: the plain
kernel vs the software-pipelined one, and the number of workgroups (default: one per 4 samples, capped at 8 192).

    python scripts/bench_interaction_bwd_variants.py            # on an MI355X"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.interaction import dot_interaction  # noqa: E402


def timed(fn, iters=30):
    for _ in range(5):
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
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    B, F, D = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 26, 16
    n = F + 1
    dense = torch.randn(B, D, device=dev, requires_grad=True)
    sparse = torch.randn(B, F * D, device=dev, requires_grad=True)
    out = dot_interaction(dense, sparse, D, True, True)
    go = torch.randn_like(out)
    by_b = 4 * B * (n * D + out.shape[1] + n * D)
    t_f = timed(lambda: dot_interaction(dense, sparse, D, True, True))

    def fb():
        o = dot_interaction(dense, sparse, D, True, True)
        o.backward(go)
        dense.grad = sparse.grad = None

    by_f = 4 * B * (n * D + out.shape[1])
    for wgs in (0, 4096, 3072, 2048, 1536, 1024):
        L.tzr_tune(b"ia_fwd_wgs", wgs)
        o2 = dot_interaction(dense, sparse, D, True, True)
        t = timed(lambda: dot_interaction(dense, sparse, D, True, True))
        print(f"B={B} forward wgs={wgs or 'auto':>5}: {t:6.1f} us = {by_f / t / 1e6:4.2f} TB/s   same output: {bool(torch.equal(o2, out))}", flush=True)
    L.tzr_tune(b"ia_fwd_wgs", 0)
    ref = None
    for plain, wgs in ((1, 0), (0, 0), (0, 8192), (0, 2048)):
        L.tzr_tune(b"ia_bwd_plain", plain)
        L.tzr_tune(b"ia_bwd_wgs", wgs)
        fb_out = dot_interaction(dense, sparse, D, True, True)
        fb_out.backward(go)
        g = sparse.grad.clone()
        dense.grad = sparse.grad = None
        if ref is None:
            ref = g
        t = timed(fb) - t_f
        print(f"B={B} plain={plain} wgs={wgs or 'auto':>5}: bwd {t:6.1f} us = {by_b / t / 1e6:4.2f} TB/s   same grads: {bool(torch.equal(g, ref))}", flush=True)
    L.tzr_tune(b"ia_bwd_plain", 0)
    L.tzr_tune(b"ia_bwd_wgs", 0)


if __name__ == "__main__":
    main()
