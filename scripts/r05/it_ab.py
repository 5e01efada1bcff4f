#!/usr/bin/env python3
"""Fused interaction kernels (csrc/interaction_top.hip) of TWO builds on one box, interleaved: the product library against
`torcheasyrec_amd/libit_old.so` (that one file of an earlier commit compiled alone:
  git show <rev>:torcheasyrec_amd/csrc/interaction_top.hip > /tmp/old/interaction_top.hip
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -I csrc -I ../include -shared /tmp/old/interaction_top.hip -o libit_old.so).
DLRM-Criteo shape (27 vectors of 16, H = 64); HIP events around each call, median of 40; outputs of the two compared bit for bit."""
import ctypes
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
    new = _lib.lib()
    old_path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libit_old.so")
    old = ctypes.CDLL(old_path)
    for name in ("tzr_dot_interaction_top_fwd", "tzr_dot_interaction_top_bwd"):
        f = getattr(old, name)
        f.restype, f.argtypes = getattr(new, name).restype, getattr(new, name).argtypes
    dev = torch.device("cuda", 0)
    D, F, H = 16, 26, 64
    n = F + 1
    width = n * (n - 1) // 2 + D * n
    st = _lib.stream_ptr(dev)
    for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,8192").split(",")]:
        torch.manual_seed(B)
        dense = torch.randn(B, D, device=dev)
        sparse = torch.randn(B, F * D, device=dev)
        W1 = torch.randn(H, width, device=dev) * 0.05
        b1 = torch.randn(H, device=dev)
        g1 = torch.randn(B, H, device=dev)
        out = {}
        for tag, L in (("old", old), ("new", new)):
            z = torch.zeros(B, width, device=dev)
            y1 = torch.zeros(B, H, device=dev)
            y1z = torch.zeros(B, H, device=dev)
            gd, gs = torch.zeros_like(dense), torch.zeros_like(sparse)

            def fwd(zp, y, L=L):
                return lambda: L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width,
                                                             _lib.ptr(b1), H, 1, zp, width, _lib.ptr(y), H, st)

            def bwd(L=L, gd=gd, gs=gs):
                L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width,
                                              None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, st)

            out[tag] = dict(fwd=fwd(None, y1), fwdz=fwd(_lib.ptr(z), y1z), bwd=bwd, t=(z, y1, y1z, gd, gs))
        for rnd in range(3):
            for tag in ("old", "new"):
                o = out[tag]
                print(f"B {B} round {rnd} {tag}: top_fwd (no z) {timed(o['fwd']):6.1f}  top_fwd+z {timed(o['fwdz']):6.1f}  top_bwd {timed(o['bwd']):6.1f} us",
                      flush=True)
        for sg in [int(x) for x in os.environ.get("IT_FWD_STAGGER", "").split(",") if x]:  # 1 half / half, 2 all row-first, 3 all row-second
            new.tzr_tune(b"it_fwd_stagger", sg)
            print(f"B {B} new, it_fwd_stagger {sg}: top_fwd (no z) {timed(out['new']['fwd']):6.1f}  top_fwd+z {timed(out['new']['fwdz']):6.1f} us", flush=True)
            torch.cuda.synchronize()
            print(f"    y1 equal to the old build's: {bool(torch.equal(out['old']['t'][1], out['new']['t'][1]))}", flush=True)
        new.tzr_tune(b"it_fwd_stagger", 0)
        out["new"]["fwd"]()
        torch.cuda.synchronize()
        same = [bool(torch.equal(a, b)) for a, b in zip(out["old"]["t"], out["new"]["t"])]
        print(f"B {B}: outputs equal bit for bit (z, y1, y1 with z, grad dense, grad sparse): {same}", flush=True)


if __name__ == "__main__":
    main()
