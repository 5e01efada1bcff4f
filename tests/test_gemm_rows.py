"""Linear layers over a tall input on the matrix cores (csrc/gemm_rows.hip: tzr_linear_rows, tzr_linear_rows_wgrad) against
torch's fp32 products -- the attention MLP of DIN on the jagged positions (/root/reference/tzrec/modules/sequence.py:101-128).
Tolerance: exact-fp32 MFMA sums in another order than torch's GEMM: <= 2e-6 of the row's / column's absolute-value product
(|x| @ |W|), the bound of a reordered fp32 sum of that length."""
import numpy as np
import pytest
import torch

from torcheasyrec_amd import _lib


def _close(got, want, scale, tol=4e-6):
    err = (got - want).abs()
    assert bool((err <= tol * scale + 1e-7).all()), float((err / (scale + 1e-30)).max())


# (K, H) pairs out of the kernel's list; N values: not a multiple of the tile, fewer rows than one tile, many turns per workgroup
@pytest.mark.parametrize("K,H", [(96, 256), (144, 256), (256, 64), (48, 64), (64, 128), (256, 256), (192, 256)])
@pytest.mark.parametrize("N,cap", [(1000, 0), (7, 0), (333, 3)])
@pytest.mark.parametrize("mode", ["bias_relu", "rowvec", "plain"])
def test_linear_rows_forward_matches_torch(dev, K, H, N, cap, mode):
    from torcheasyrec_amd.dense import linear_rows, linear_rows_supported

    g = torch.Generator().manual_seed(K * 1000 + H + N)
    x = torch.randn(N, K + 8, generator=g)[:, :K + 8]
    W = torch.randn(H, K, generator=g) / K ** 0.5
    b = torch.randn(H, generator=g)
    R = 5
    rv = torch.randn(R, H, generator=g)
    idx = torch.randint(0, R, (N,), generator=g, dtype=torch.int32)
    xd = x.to(dev)
    assert linear_rows_supported(xd, K, H)
    if mode == "rowvec" and not _lib.lib().tzr_linear_rows_supported(K, H) & 2:
        L = _lib.lib()  # (shapes whose row-vector form is not built refuse it)
        assert L.tzr_linear_rows(_lib.ptr(xd), xd.stride(0), _lib.ptr(W.to(dev)), K, 1, None, _lib.ptr(rv.to(dev)), H, _lib.ptr(idx.to(dev)), 1, N, K,
                                 H, _lib.ptr(torch.empty(N, H).to(dev)), H, _lib.stream_ptr(dev)) == -4
        return
    if cap:
        _lib.lib().tzr_tune(b"gemm_rows_wg", cap)
    try:
        if mode == "bias_relu":
            got = linear_rows(xd, W.to(dev), b.to(dev), relu=True, K=K)
            want = torch.relu(x[:, :K].double() @ W.double().t() + b.double())
        elif mode == "rowvec":
            got = linear_rows(xd, W.to(dev), None, relu=True, rowvec=rv.to(dev), row_index=idx.to(dev), K=K)
            want = torch.relu(x[:, :K].double() @ W.double().t() + rv.double()[idx.long()])
        else:
            got = linear_rows(xd, W.to(dev), K=K)
            want = x[:, :K].double() @ W.double().t()
        again = linear_rows(xd, W.to(dev), b.to(dev), relu=True, K=K) if mode == "bias_relu" else None
    finally:
        _lib.lib().tzr_tune(b"gemm_rows_wg", 0)
    scale = x[:, :K].abs().double() @ W.abs().double().t() + 1.0
    _close(got.cpu().double(), want, scale)
    if again is not None:
        assert torch.equal(again.cpu(), got.cpu())


@pytest.mark.parametrize("K,H", [(256, 96), (256, 48), (64, 144), (128, 192), (256, 144), (256, 192)])
@pytest.mark.parametrize("N", [500, 16])
def test_linear_rows_input_gradient_matches_torch(dev, K, H, N):
    """g [N, K] @ weight [K, H] (the weight of a layer K <- H as nn.Linear stores it: no transpose, no copy)"""
    from torcheasyrec_amd.dense import linear_rows, linear_rows_supported

    gen = torch.Generator().manual_seed(K + 7 * H + N)
    g = torch.randn(N, K, generator=gen)
    W = torch.randn(K, H, generator=gen) / K ** 0.5
    assert linear_rows_supported(g.to(dev), K, H)
    got = linear_rows(g.to(dev), W.to(dev), out_major=False)
    want = g.double() @ W.double()
    _close(got.cpu().double(), want, g.abs().double() @ W.abs().double() + 1.0)


@pytest.mark.parametrize("H,K", [(256, 96), (256, 144), (64, 256), (256, 48), (128, 128), (64, 48)])
@pytest.mark.parametrize("N,cap", [(900, 0), (5, 0), (700, 2)])
def test_linear_rows_weight_gradient_matches_torch(dev, H, K, N, cap):
    from torcheasyrec_amd.dense import linear_rows_wgrad, linear_rows_wgrad_supported

    gen = torch.Generator().manual_seed(H + 3 * K + N)
    g = torch.randn(N, H, generator=gen)
    x = torch.randn(N, K + 4, generator=gen)
    assert linear_rows_wgrad_supported(g.to(dev), x.to(dev), K)
    if cap:
        _lib.lib().tzr_tune(b"gemm_rows_wg", cap)
    try:
        got = linear_rows_wgrad(g.to(dev), x.to(dev), K)
        again = linear_rows_wgrad(g.to(dev), x.to(dev), K)
    finally:
        _lib.lib().tzr_tune(b"gemm_rows_wg", 0)
    want = g.double().t() @ x[:, :K].double()
    _close(got.cpu().double(), want, g.abs().double().t() @ x[:, :K].abs().double() + 1.0)
    assert torch.equal(again.cpu(), got.cpu())  # no float atomics: bit-reproducible


def test_unsupported_shapes_are_refused(dev):
    from torcheasyrec_amd.dense import linear_rows_supported, linear_rows_wgrad_supported

    x = torch.randn(64, 100).to(dev)
    assert not linear_rows_supported(x, 100, 64)
    assert not linear_rows_supported(x, 96, 80)
    assert not linear_rows_wgrad_supported(torch.randn(64, 80).to(dev), x, 96)
    L = _lib.lib()
    assert L.tzr_linear_rows(_lib.ptr(x), 100, _lib.ptr(x), 100, 1, None, None, 0, None, 0, 64, 100, 64, _lib.ptr(x), 100, None) == -4  # TZR_ERR_UNSUPPORTED
