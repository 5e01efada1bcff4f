#!/usr/bin/env python3
"""Fused interaction + first top layer (csrc/interaction_top.hip) against the unfused ops, kernel by kernel, at the
DLRM-Criteo shape (n = 27 rows of 16, H = 64): HIP events around each call, a GPU-side sleep first."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.dense import weight_grad  # noqa: E402


def timed(fn, iters=20):
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
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench

    bench.enable_tunable_gemm()  # the unfused GEMMs as the step runs them
    # IT_LIB: a variant build (scripts/r06/build_variant.py) instead of the product library
    _lib.use_library(os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), os.environ["IT_LIB"]) if os.environ.get("IT_LIB") else _build.build())
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    D, F, H = 16, 26, 64
    n = F + 1
    width = n * (n - 1) // 2 + D * n
    st = _lib.stream_ptr(dev)
    for B in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,8192").split(",")]:
        dense = torch.randn(B, D, device=dev)
        sparse = torch.randn(B, F * D, device=dev)
        W1 = torch.randn(H, width, device=dev) * 0.05
        b1 = torch.randn(H, device=dev)
        g1 = torch.randn(B, H, device=dev)
        z = torch.empty(B, width, device=dev)
        dz = torch.empty(B, width, device=dev)
        y1 = torch.empty(B, H, device=dev)
        gd, gs = torch.empty_like(dense), torch.empty_like(sparse)
        W1t = W1.t()

        def ia_fwd():
            L.tzr_dot_interaction_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(z), width, 1, 1, st)

        def gemm_fwd():
            torch._addmm_activation(b1, z, W1t, use_gelu=False)

        def gemm_dz():
            torch.mm(g1, W1, out=dz)

        def ia_bwd():
            L.tzr_dot_interaction_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(dz), width, 1, 1, _lib.ptr(gd), D,
                                      _lib.ptr(gs), F * D, st)

        def top_fwd(zp):
            return lambda: L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width,
                                                         _lib.ptr(b1), H, 1, zp, width, _lib.ptr(y1), H, st)

        def top_bwd():
            L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width,
                                          None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, st)

        print(f"B {B}: unfused  ia_fwd {timed(ia_fwd):6.1f}  gemm_fwd {timed(gemm_fwd):6.1f}  gemm_dz {timed(gemm_dz):6.1f}  "
              f"ia_bwd {timed(ia_bwd):6.1f}  dW {timed(lambda: weight_grad(g1, z)):6.1f} us", flush=True)
        for wgs in [int(x) for x in os.environ.get("IT_WGS", "0").split(",")]:
            L.tzr_tune(b"it_wgs", wgs)
            print(f"B {B}: fused (it_wgs {wgs:4d})  top_fwd+z {timed(top_fwd(_lib.ptr(z))):6.1f}  top_fwd (no z) {timed(top_fwd(None)):6.1f}  "
                  f"top_bwd {timed(top_bwd):6.1f} us", flush=True)
        L.tzr_tune(b"it_wgs", 0)
        ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, 1, H), dev)
        dW = torch.empty(H, width, device=dev)

        def top_wgrad():
            L.tzr_dot_interaction_top_wgrad(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, None, _lib.ptr(dW),
                                            width, _lib.ptr(ws), ws.numel(), st)

        t_w = timed(top_wgrad)
        L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1,
                                      _lib.ptr(z), width, _lib.ptr(y1), H, st)
        ref = weight_grad(g1, z)
        err = float((dW - ref).abs().max()) / float(ref.abs().max())
        print(f"B {B}: top_wgrad (z rebuilt on the chip, kernel + reduce) {t_w:6.1f} us; max |dW - g1^T z| / max |.| = {err:.2e}", flush=True)
        for dbg in [int(x) for x in os.environ.get("WG_DEBUG", "").split(",") if x]:
            L.tzr_tune(b"wg_debug", dbg)
            print(f"B {B}: wg_debug {dbg} (1 no product, 2 no tile build, 4 no loads): top_wgrad {timed(top_wgrad):6.1f} us", flush=True)
        L.tzr_tune(b"wg_debug", 0)
        for sg in [int(x) for x in os.environ.get("IT_STAGGER", "").split(",") if x]:
            L.tzr_tune(b"it_stagger", sg)
            print(f"B {B}: it_stagger {sg}: top_bwd {timed(top_bwd):6.1f} us", flush=True)
        L.tzr_tune(b"it_stagger", 0)
        for sg in [int(x) for x in os.environ.get("IT_FWD_STAGGER", "").split(",") if x]:  # 1 half / half, 2 all row-first, 3 all row-second
            L.tzr_tune(b"it_fwd_stagger", sg)
            print(f"B {B}: it_fwd_stagger {sg}: top_fwd+z {timed(top_fwd(_lib.ptr(z))):6.1f}  top_fwd (no z) {timed(top_fwd(None)):6.1f} us", flush=True)
        L.tzr_tune(b"it_fwd_stagger", 0)
        for dbg in [int(x) for x in os.environ.get("IT_DEBUG", "").split(",") if x]:
            if L.tzr_tune(b"it_debug", dbg) != 0:
                break  # (the phase-skipping knob existed in the first versions only: profiles/r03ae)
            print(f"B {B}: it_debug {dbg}: top_fwd+z {timed(top_fwd(_lib.ptr(z))):6.1f}  top_fwd (no z) {timed(top_fwd(None)):6.1f}  "
                  f"top_bwd {timed(top_bwd):6.1f} us", flush=True)
        L.tzr_tune(b"it_debug", 0)


def prof():
    """In-kernel phase times: a second library built with -DIT_PROF (scripts/build_prof_lib.sh, in-tree so that it
    travels to the GPU box) whose fused kernels sum, per wave, the shader clocks spent in each phase of the tile loop."""
    import ctypes

    import numpy as np

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torcheasyrec_amd", "libtzrec_hip_prof.so")
    _lib.use_library(path)
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    D, F, H, B = 16, 26, 64, 65536
    width = 27 * 26 // 2 + D * 27
    st = _lib.stream_ptr(dev)
    dense, sparse = torch.randn(B, D, device=dev), torch.randn(B, F * D, device=dev)
    W1, b1, g1 = torch.randn(H, width, device=dev) * 0.05, torch.randn(H, device=dev), torch.randn(B, H, device=dev)
    z, y1 = torch.empty(B, width, device=dev), torch.empty(B, H, device=dev)
    gd, gs = torch.empty_like(dense), torch.empty_like(sparse)
    tab = torch.zeros(256 * 16 * 6, dtype=torch.int64, device=dev)
    L.tzr_it_prof_table(ctypes.c_void_p(tab.data_ptr()))
    ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, 1, H), dev)
    dW = torch.empty(H, width, device=dev)
    names = {"wgrad": ["load wait", "X stores + MFMAs", "load issue", "barrier (L)", "pair stores (L) / product (M)", "barrier (M)"],
             "bwd": ["wait", "x-image", "product(first)", "contract", "pt+stores", "product(second)"],
             "fwd+z": ["wait1", "row(first)", "product", "row(second)", "wait2", "reduce"],
             "fwd": ["wait1", "row(first)", "product", "row(second)", "wait2", "reduce"]}
    kinds = [k for k in os.environ.get("PROF_KINDS", "wgrad,bwd,fwd+z,fwd").split(",") if k]
    L.tzr_tune(b"wg_debug", int(os.environ.get("PROF_WG_DEBUG", "0")))
    for kind in kinds:
        for it in range(3):
            tab.zero_()
            if kind == "wgrad":
                L.tzr_dot_interaction_top_wgrad(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, None, _lib.ptr(dW),
                                                width, _lib.ptr(ws), ws.numel(), st)
            elif kind == "bwd":
                L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width,
                                              None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, st)
            else:
                L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1,
                                              _lib.ptr(z) if kind == "fwd+z" else None, width, _lib.ptr(y1), H, st)
            torch.cuda.synchronize()
        per_wg = 32.0 if kind == "wgrad" else 16.0  # tiles per workgroup
        t = tab.cpu().numpy().reshape(256, 16, 6).astype(np.float64) / per_wg
        print(f"{kind}: shader clocks per tile, mean over workgroups; per wave (rows) x phase (cols) {names[kind]}")
        if kind == "wgrad":  # by column group (workgroup b: group (b >> 3) & 3)
            for gidx in range(4):
                sel = [b for b in range(256) if ((b >> 3) & 3) == gidx]
                mg = t[sel].mean(axis=0)
                print(f"  group {gidx}: loaders (waves 0-7) " + " ".join(f"{v:8.0f}" for v in mg[:8].mean(axis=0)[:5])
                      + "   multipliers (waves 8-15) " + " ".join(f"{v:8.0f}" for v in mg[8:].mean(axis=0)[4:]))
        m = t.mean(axis=0)
        for w in range(16):
            print(f"  wave {w:2d}: " + " ".join(f"{v:8.0f}" for v in m[w]) + f"   sum {m[w].sum():8.0f}")
        print("  all    : " + " ".join(f"{v:8.0f}" for v in m.mean(axis=0)) + f"   sum {m.mean(axis=0).sum():8.0f}", flush=True)


if __name__ == "__main__":
    if "--prof" in sys.argv:
        sys.argv.remove("--prof")
        prof()
    else:
        main()
