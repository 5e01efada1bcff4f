#!/usr/bin/env python3
"""The 33-64-row dot-interaction backward (tzr_dot_interaction_bwd) timed through the C ABI at B = 65 536, per
tzr_tune("ia_gen_wgs") grid: time, algorithmic bytes (grad_out + X read, dX written), TB/s.

    python scripts/bench_interaction_gen.py [wgs ...]          # on an MI355X
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402


def main():
    _lib.use_native()
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    B = 65536
    grids = [int(a) for a in sys.argv[1:]] or [0]
    shapes = ((39, 16), (47, 32), (63, 32), (63, 64))
    if os.environ.get("IA_GEN_SMALL"):
        shapes = ((26, 32), (26, 64), (16, 64), (31, 8))
    if os.environ.get("IA_GEN_SHAPES"):  # "39x16,8x128": F x D
        shapes = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["IA_GEN_SHAPES"].split(","))
    for F, D in shapes:
        n = F + 1
        width = n * (n - 1) // 2 + n * D
        dense = torch.randn(B, D, device=dev)
        sparse = torch.randn(B, F * D, device=dev)
        gout = torch.randn(B, width, device=dev)
        gd, gs = torch.empty_like(dense), torch.empty_like(sparse)
        st = _lib.stream_ptr(dev)

        def run():
            rc = L.tzr_dot_interaction_bwd(_lib.ptr(dense), dense.stride(0), _lib.ptr(sparse), sparse.stride(0), F, D, B,
                                           _lib.ptr(gout), gout.stride(0), 1, 1, _lib.ptr(gd), gd.stride(0), _lib.ptr(gs),
                                           gs.stride(0), st)
            _lib.check(rc, "tzr_dot_interaction_bwd")

        by = 4 * B * (width + 2 * n * D)
        for wgs in grids:
            L.tzr_tune(b"ia_gen_wgs", wgs)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                run()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 20 * 1e3
            print(f"n={n:3d} D={D:3d} wgs={wgs:5d}  bwd {us:7.1f} us = {by / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
