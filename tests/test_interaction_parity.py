"""K9 dot interaction and K10 FM against the oracle restatement of the reference modules
(/root/reference/tzrec/modules/interaction.py:57-91, fm.py:17-42).  fp32, 1e-5 relative."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd.interaction import FactorizationMachine, InteractionArch, dot_interaction  # noqa: E402

RTOL, ATOL = 1e-5, 1e-5


@pytest.mark.parametrize("N", [2, 4, 16, 17, 27, 32])
@pytest.mark.parametrize("B", [1, 5, 70])
@pytest.mark.parametrize("plain", [0, 1])
def test_interaction_arch_forward_backward(dev, N, B, plain):
    from torcheasyrec_amd import _lib

    _lib.lib().tzr_tune(b"ia_bwd_plain", plain)
    _lib.lib().tzr_tune(b"ia_bwd_wgs", 0 if plain else 2)  # pipelined: two workgroups walk the whole batch
    _lib.lib().tzr_tune(b"ia_fwd_wgs", 0 if plain else 3)
    g = torch.Generator().manual_seed(N * 1000 + B)
    x = torch.randn(B, N, 16, generator=g)
    # asymmetric rows so a transposed fragment map cannot pass
    x = x * torch.linspace(0.5, 1.5, N).view(1, N, 1)
    m = InteractionArch(N)
    assert m.output_dim() == N * (N - 1) // 2
    xd = x.clone().to(dev).requires_grad_(True)
    out = m(xd)
    xr = x.clone().requires_grad_(True)
    ref = orc.dot_interaction(xr)
    assert out.shape == ref.shape
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
    go = torch.randn(ref.shape, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("N,D,B", [(5, 8, 3), (27, 32, 4), (32, 20, 3), (33, 4, 3), (40, 16, 3), (48, 8, 5), (49, 8, 3), (9, 128, 2), (64, 64, 2), (64, 12, 5),
                                   (2, 4, 70), (70, 8, 2), (100, 32, 1)])
def test_interaction_arch_general_shapes(dev, N, D, B):
    """Shapes outside the D = 16, n <= 32 specialisation: generalised MFMA kernels up to 64 rows (2 or
    4 row blocks, partial 16-column blocks), the LDS/VALU kernel beyond."""
    g = torch.Generator().manual_seed(N * 131 + D)
    x = torch.randn(B, N, D, generator=g) * torch.linspace(0.5, 1.5, N).view(1, N, 1)
    xd = x.clone().to(dev).requires_grad_(True)
    out = InteractionArch(N)(xd)
    xr = x.clone().requires_grad_(True)
    ref = orc.dot_interaction(xr)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL * D / 16)
    go = torch.randn(ref.shape, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=RTOL, atol=ATOL * max(1, N // 16))


@pytest.mark.parametrize("cat_dense,cat_sparse", [(True, True), (False, True), (True, False)])
def test_fused_dlrm_interaction_general_dim(dev, cat_dense, cat_sparse):
    """the DLRM concatenation with embedding_dim = 32 (general kernel)"""
    B, F, D = 6, 11, 32
    g = torch.Generator().manual_seed(17)
    dense = torch.randn(B, D, generator=g)
    sparse = torch.randn(B, F * D, generator=g)
    dd = dense.clone().to(dev).requires_grad_(True)
    sd = sparse.clone().to(dev).requires_grad_(True)
    out = dot_interaction(dd, sd, D, cat_dense, cat_sparse)
    dr = dense.clone().requires_grad_(True)
    sr = sparse.clone().requires_grad_(True)
    parts = [orc.dot_interaction(torch.cat([dr.unsqueeze(1), sr.reshape(B, F, D)], dim=1))]
    if cat_dense:
        parts.append(dr)
    if cat_sparse:
        parts.append(sr)
    ref = torch.cat(parts, dim=-1)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=2 * ATOL)
    go = torch.randn(ref.shape, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    torch.testing.assert_close(dd.grad.cpu(), dr.grad, rtol=RTOL, atol=2 * ATOL)
    torch.testing.assert_close(sd.grad.cpu(), sr.grad, rtol=RTOL, atol=2 * ATOL)


def test_interaction_shape_beyond_lds_is_refused(dev):
    x = torch.randn(1, 200, 128).to(dev)
    with pytest.raises(RuntimeError, match="UNSUPPORTED|unsupported|-3|tzr_dot_interaction"):
        InteractionArch(200)(x)


def test_shape_fixture_from_reference(dev):
    """tzrec/modules/interaction_test.py:47-54: feature_num=4 -> output (10, 6)."""
    m = InteractionArch(4)
    out = m(torch.randn(10, 4, 16).to(dev))
    assert tuple(out.shape) == (10, 6)


@pytest.fixture(params=[(0, 0), (1, 0), (0, 3), (1, 2)], ids=["default", "plain", "pipelined-3wg", "plain-2wg"])
def ia_bwd_variant(request):
    """tzr_tune knobs of the D = 16 backward: the software-pipelined kernel (default) or the plain one, a grid smaller than the batch
    (a wave then walks several samples: loop carried state, the tail sample that does not exist)"""
    from torcheasyrec_amd import _lib

    pipe, wgs = request.param
    yield lambda: (_lib.lib().tzr_tune(b"ia_bwd_plain", pipe), _lib.lib().tzr_tune(b"ia_bwd_wgs", wgs),
                   _lib.lib().tzr_tune(b"ia_fwd_wgs", wgs))
    if _lib._lib is not None:
        _lib.lib().tzr_tune(b"ia_bwd_plain", 0)
        _lib.lib().tzr_tune(b"ia_bwd_wgs", 0)


@pytest.mark.parametrize("cat_dense,cat_sparse", [(True, True), (True, False), (False, True), (False, False)])
def test_fused_dlrm_interaction(dev, cat_dense, cat_sparse, ia_bwd_variant):
    """[interactions | dense | sparse] exactly as DLRM.predict concatenates it
    (/root/reference/tzrec/models/dlrm.py:123-130)."""
    ia_bwd_variant()
    B, F, D = 77, 26, 16
    g = torch.Generator().manual_seed(3)
    dense = torch.randn(B, D, generator=g)
    sparse = torch.randn(B, F * D, generator=g)
    dd = dense.clone().to(dev).requires_grad_(True)
    sd = sparse.clone().to(dev).requires_grad_(True)
    out = dot_interaction(dd, sd, D, cat_dense, cat_sparse)
    dr = dense.clone().requires_grad_(True)
    sr = sparse.clone().requires_grad_(True)
    feat = torch.cat([dr.unsqueeze(1), sr.reshape(B, F, D)], dim=1)
    parts = [orc.dot_interaction(feat)]
    if cat_dense:
        parts.append(dr)
    if cat_sparse:
        parts.append(sr)
    ref = torch.cat(parts, dim=-1)
    assert out.shape == ref.shape
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
    if cat_dense and cat_sparse:
        assert out.shape[1] == 783  # 351 + 16 + 416, the DLRM-Criteo final-MLP input
        assert torch.equal(out.detach().cpu()[:, 351:367], dense)  # pass-through is a copy
        assert torch.equal(out.detach().cpu()[:, 367:], sparse)
    go = torch.randn(ref.shape, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    torch.testing.assert_close(dd.grad.cpu(), dr.grad, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(sd.grad.cpu(), sr.grad, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("B,F,D", [(4, 26, 16), (1, 3, 4), (300, 7, 32), (33, 2, 8)])
def test_fm(dev, B, F, D):
    g = torch.Generator().manual_seed(B + F + D)
    x = torch.randn(B, F, D, generator=g)
    xd = x.clone().to(dev).requires_grad_(True)
    out = FactorizationMachine()(xd)
    xr = x.clone().requires_grad_(True)
    ref = orc.fm(xr)
    assert tuple(out.shape) == (B, D)  # fm_test.py:23-32 asserts (4, 16)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=RTOL, atol=ATOL)
    go = torch.randn(B, D, generator=g)
    out.backward(go.to(dev))
    ref.backward(go)
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("F,D,B,wgs", [(39, 16, 7, 0), (47, 24, 9, 1), (63, 32, 5, 2), (63, 32, 9, 1), (48, 8, 3, 0), (20, 40, 11, 2), (8, 128, 5, 1), (15, 24, 9, 2), (1, 8, 6, 0)])
def test_fused_dlrm_interaction_persistent_rows(dev, F, D, B, wgs):
    """The generalised MFMA backward (D != 16 or 33-64 rows) behind the DLRM concatenation: batched pair-gradient
    loads, pass-through gradients, the next sample's loads in flight (49-64 rows), and a grid smaller than the
    batch (tzr_tune "ia_gen_wgs": one or two workgroups walk every sample)."""
    from torcheasyrec_amd import _lib

    _lib.lib().tzr_tune(b"ia_gen_wgs", wgs)
    try:
        for cat_dense, cat_sparse in ((True, True), (False, True), (False, False)):
            g = torch.Generator().manual_seed(F + D)
            dense = torch.randn(B, D, generator=g)
            sparse = torch.randn(B, F * D, generator=g)
            dd = dense.clone().to(dev).requires_grad_(True)
            sd = sparse.clone().to(dev).requires_grad_(True)
            out = dot_interaction(dd, sd, D, cat_dense, cat_sparse)
            dr = dense.clone().requires_grad_(True)
            sr = sparse.clone().requires_grad_(True)
            parts = [orc.dot_interaction(torch.cat([dr.unsqueeze(1), sr.reshape(B, F, D)], dim=1))]
            if cat_dense:
                parts.append(dr)
            if cat_sparse:
                parts.append(sr)
            ref = torch.cat(parts, dim=-1)
            go = torch.randn(ref.shape, generator=g)
            out.backward(go.to(dev))
            ref.backward(go)
            torch.testing.assert_close(dd.grad.cpu(), dr.grad, rtol=RTOL, atol=4 * ATOL)
            torch.testing.assert_close(sd.grad.cpu(), sr.grad, rtol=RTOL, atol=4 * ATOL)
    finally:
        _lib.lib().tzr_tune(b"ia_gen_wgs", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("F,D", [(39, 16), (63, 32), (26, 32), (8, 128)])
def test_persistent_backward_walks_a_large_batch(F, D):
    """B = 9 001 on the chip: the resident grid of the generalised backward (768 / 1 024 workgroups) takes two to five
    samples per wave, the last round is partial, and (49-64 rows) the next sample's loads are in flight across the
    stores.  Against torch autograd in fp64 on the same inputs, 1e-5 of the largest entry; a second launch is bit-identical."""
    from torcheasyrec_amd import _lib

    _lib.use_native()
    dev = torch.device("cuda", 0)
    B = 9001
    g = torch.Generator(device="cpu").manual_seed(F * 7 + D)
    dense = torch.randn(B, D, generator=g).to(dev).requires_grad_(True)
    sparse = torch.randn(B, F * D, generator=g).to(dev).requires_grad_(True)
    out = dot_interaction(dense, sparse, D, True, True)
    go = torch.randn(out.shape, generator=g).to(dev)
    gd, gs = torch.autograd.grad(out, (dense, sparse), go, retain_graph=True)
    gd2, gs2 = torch.autograd.grad(out, (dense, sparse), go)
    assert torch.equal(gd, gd2) and torch.equal(gs, gs2)
    d64 = dense.detach().double().requires_grad_(True)
    s64 = sparse.detach().double().requires_grad_(True)
    x = torch.cat([d64.unsqueeze(1), s64.reshape(B, F, D)], dim=1)
    iu = torch.triu_indices(F + 1, F + 1, 1, device=dev)
    ref = torch.cat([torch.bmm(x, x.transpose(1, 2))[:, iu[0], iu[1]], d64, s64], dim=1)
    torch.testing.assert_close(out.detach().double(), ref.detach(), rtol=1e-5, atol=1e-5 * float(ref.abs().max()))
    rd, rs = torch.autograd.grad(ref, (d64, s64), go.double())
    torch.testing.assert_close(gd.double(), rd, rtol=1e-5, atol=1e-5 * float(rd.abs().max()))
    torch.testing.assert_close(gs.double(), rs, rtol=1e-5, atol=1e-5 * float(rs.abs().max()))


def test_every_row_count_up_to_64_on_the_emulator(emu_path):
    """n = 2 .. 64 x D in {4, 12, 24, 40} x (dense row / concatenation forms) x grid knob: every row-block count, every
    partial 16-column block, the run-time S pitch of 33-48 rows, all three persistent-grid forms of the backward.  Emulator
    only (1 008 cases in ~10 s on the host; on the chip the parametrised tests above cover the shape classes)."""
    from torcheasyrec_amd import _lib

    _lib.use_library(emu_path)  # (the `dev` fixture of the other tests selects its library itself)
    try:
        bad = []
        for n in range(2, 65):
            for D in (4, 12, 24, 40):
                for hd, cd, cs, wgs in ((1, True, True, 0), (1, False, True, 1), (0, False, False, 2), (1, True, False, 1)):
                    F, B = n - hd, 5
                    g = torch.Generator().manual_seed(n * 100 + D)
                    dense = torch.randn(B, D, generator=g) if hd else None
                    sparse = torch.randn(B, F * D, generator=g)
                    _lib.lib().tzr_tune(b"ia_gen_wgs", wgs)
                    dd = dense.clone().requires_grad_(True) if hd else None
                    sd = sparse.clone().requires_grad_(True)
                    out = dot_interaction(dd, sd, D, cd, cs)
                    dr = dense.clone().requires_grad_(True) if hd else None
                    sr = sparse.clone().requires_grad_(True)
                    rows = ([dr.unsqueeze(1)] if hd else []) + [sr.reshape(B, F, D)]
                    parts = [orc.dot_interaction(torch.cat(rows, dim=1))]
                    if cd and hd:
                        parts.append(dr)
                    if cs:
                        parts.append(sr)
                    ref = torch.cat(parts, dim=-1)
                    go = torch.randn(ref.shape, generator=g)
                    out.backward(go)
                    ref.backward(go)
                    ok = (torch.allclose(out.detach(), ref.detach(), rtol=1e-5, atol=1e-4) and torch.allclose(sd.grad, sr.grad, rtol=1e-5, atol=1e-4)
                          and (not hd or torch.allclose(dd.grad, dr.grad, rtol=1e-5, atol=1e-4)))
                    if not ok:
                        bad.append((n, D, hd, cd, cs, wgs))
        assert not bad, bad[:10]
    finally:
        _lib.lib().tzr_tune(b"ia_gen_wgs", 0)
