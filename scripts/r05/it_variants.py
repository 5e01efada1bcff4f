#!/usr/bin/env python3
"""Build variants of the fused interaction kernels on one box, interleaved: every `torcheasyrec_amd/libit_x_<tag>.so`
(interaction_top.hip + interaction_wgrad.hip compiled alone with an experiment's -D flags:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I csrc -I ../include -D... -shared \
        csrc/interaction_top.hip csrc/interaction_wgrad.hip -o libit_x_<tag>.so)
timed at the DLRM-Criteo shape (27 vectors of 16, H = 64, B = 65 536): forward without z, backward, weight gradient;
HIP events around each call, median of 40, three rounds; outputs compared with the first variant's bit for bit."""
import ctypes
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402


def timed(fn, iters=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return v[len(v) // 2]


def main():
    _lib.use_library(_build.build())
    ref = _lib.lib()
    here = os.path.dirname(_lib.LIB_PATH)
    paths = sorted(glob.glob(os.path.join(here, "libit_x_*.so")))
    want = [t for t in os.environ.get("IT_VARIANTS", "").split(",") if t]
    libs = []
    for pth in paths:
        tag = os.path.basename(pth)[len("libit_x_"):-3]
        if want and tag not in want:
            continue
        L = ctypes.CDLL(pth)
        for name in ("tzr_dot_interaction_top_fwd", "tzr_dot_interaction_top_bwd", "tzr_dot_interaction_top_wgrad",
                     "tzr_dot_interaction_top_wgrad_workspace"):
            f = getattr(L, name)
            f.restype, f.argtypes = getattr(ref, name).restype, getattr(ref, name).argtypes
        libs.append((tag, L))
    libs.sort(key=lambda x: (x[0] != "base", x[0]))
    dev = torch.device("cuda", 0)
    D, F, H = 16, 26, 64
    n = F + 1
    width = n * (n - 1) // 2 + D * n
    st = _lib.stream_ptr(dev)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    torch.manual_seed(B)
    dense = torch.randn(B, D, device=dev)
    sparse = torch.randn(B, F * D, device=dev)
    W1 = torch.randn(H, width, device=dev) * 0.05
    b1 = torch.randn(H, device=dev)
    g1 = torch.randn(B, H, device=dev)
    runs = {}
    for tag, L in libs:
        y1 = torch.zeros(B, H, device=dev)
        gd, gs = torch.zeros_like(dense), torch.zeros_like(sparse)
        dW = torch.zeros(H, width, device=dev)
        ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, 1, H), dev)

        def fwd(L=L, y1=y1):
            L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1,
                                          None, width, _lib.ptr(y1), H, st)

        def bwd(L=L, gd=gd, gs=gs):
            L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width,
                                          None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, st)

        def wgrad(L=L, dW=dW, ws=ws):
            L.tzr_dot_interaction_top_wgrad(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, None, _lib.ptr(dW),
                                            width, _lib.ptr(ws), ws.numel(), st)

        runs[tag] = dict(fwd=fwd, bwd=bwd, wgrad=wgrad, t=(y1, gd, gs, dW))
    for rnd in range(3):
        for tag, _ in libs:
            o = runs[tag]
            print(f"B {B} round {rnd} {tag:>8s}: top_fwd (no z) {timed(o['fwd']):6.1f}  top_bwd {timed(o['bwd']):6.1f}  top_wgrad {timed(o['wgrad']):6.1f} us",
                  flush=True)
    first = libs[0][0]
    for tag, _ in libs[1:]:
        same = [bool(torch.equal(a, b)) for a, b in zip(runs[first]["t"], runs[tag]["t"])]
        print(f"{tag} == {first} bit for bit (y1, grad dense, grad sparse, dW): {same}", flush=True)


if __name__ == "__main__":
    main()
