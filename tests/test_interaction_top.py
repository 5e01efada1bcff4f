"""The dot interaction fused with the first layer behind it (csrc/interaction_top.hip) against the unfused ops
(tzr_dot_interaction_fwd / bwd + torch GEMMs) and against plain torch autograd.  fp32 MFMA in another summation
order than the GEMM library: forward to 2e-6 of the largest entry, gradients to 1e-5 of the largest entry."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

from torcheasyrec_amd import _lib  # noqa: E402


def _close(a, b, rtol):
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= rtol * scale + 1e-9, (float((a - b).abs().max()), scale)


def _torch_z(dense, sparse, D):
    B = sparse.shape[0]
    X = torch.cat([dense.unsqueeze(1), sparse.view(B, -1, D)], dim=1)
    n = X.shape[1]
    iu = torch.triu_indices(n, n, offset=1, device=X.device)
    pairs = torch.bmm(X, X.transpose(1, 2))[:, iu[0], iu[1]]
    return torch.cat([pairs, dense, sparse], dim=1)


@pytest.mark.parametrize("B,F", [(1, 26), (16, 26), (37, 26), (600, 26), (50, 3), (33, 1), (40, 28), (21, 15), (35, 31)])
def test_top_fwd_bwd_match_torch(dev, B, F):
    D, H = 16, 64
    torch.manual_seed(B * 31 + F)
    L = _lib.lib()
    assert L.tzr_dot_interaction_top_supported(F, D, 1, H) == 1
    dense = torch.randn(B, D, device=dev, requires_grad=True)
    sparse = torch.randn(B, F * D, device=dev, requires_grad=True)
    n = F + 1
    width = n * (n - 1) // 2 + D * n
    lin = torch.nn.Linear(width, H).to(dev)
    g1 = torch.randn(B, H, device=dev)
    scale = torch.tensor([0.5], device=dev)
    z_ref = _torch_z(dense, sparse, D)
    pre = lin(z_ref)
    (pre * g1 * 0.5).sum().backward()
    stream = _lib.stream_ptr(sparse.device)
    for with_z in (True, False):
        z = torch.full((B, width), float("nan"), device=dev) if with_z else None
        y1 = torch.empty(B, H, device=dev)
        _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(lin.weight),
                                                 width, _lib.ptr(lin.bias), H, 1, _lib.ptr(z), width, _lib.ptr(y1), H, stream), "fwd")
        _close(y1, torch.relu(pre).detach(), 2e-6)
        if with_z:
            _close(z, z_ref.detach(), 1e-6)
    ypre = torch.empty(B, H, device=dev)
    _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(lin.weight), width,
                                             None, H, 0, None, 0, _lib.ptr(ypre), H, stream), "fwd")
    _close(ypre, (pre - lin.bias).detach(), 2e-6)
    gd, gs = torch.full_like(dense, float("nan")), torch.full_like(sparse, float("nan"))
    _lib.check(L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H,
                                             _lib.ptr(lin.weight), width, _lib.ptr(scale), _lib.ptr(gd), D, _lib.ptr(gs), F * D,
                                             stream), "bwd")
    _close(gd, dense.grad, 1e-5)
    _close(gs, sparse.grad, 1e-5)


def _wgrad(dense, sparse, F, D, g1, scale, ldw=None):
    L = _lib.lib()
    B, H = sparse.shape[0], g1.shape[1]
    hd = dense is not None
    n = F + (1 if hd else 0)
    width = n * (n - 1) // 2 + D * n
    ldw = ldw or width
    dW = torch.full((H, ldw), float("nan"), device=sparse.device)
    ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, int(hd), H), sparse.device)
    _lib.check(L.tzr_dot_interaction_top_wgrad(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), g1.stride(0), H,
                                               _lib.ptr(scale), _lib.ptr(dW), ldw, _lib.ptr(ws), ws.numel(),
                                               _lib.stream_ptr(sparse.device)), "wgrad")
    return dW, width


@pytest.mark.parametrize("B,F", [(1, 26), (16, 26), (37, 26), (600, 26), (2100, 26), (50, 3), (33, 1), (40, 28), (21, 15), (70, 16), (35, 31)])
def test_top_wgrad_matches_torch(dev, B, F):
    """dW1 = scale * g1^T z with z rebuilt on the chip (csrc/interaction_wgrad.hip) against autograd's weight gradient of the
    Linear over the materialised z: to 1e-5 of the largest entry; twice the same bits; columns behind the width untouched."""
    D, H = 16, 64
    torch.manual_seed(B * 17 + F)
    dense = torch.randn(B, D, device=dev)
    sparse = torch.randn(B, F * D, device=dev)
    g1 = torch.randn(B, H, device=dev)
    scale = torch.tensor([0.5], device=dev)
    z = _torch_z(dense, sparse, D).double()
    ref = (0.5 * (g1.double().t() @ z)).float()
    dW, width = _wgrad(dense, sparse, F, D, g1, scale, ldw=_torch_z(dense, sparse, D).shape[1] + 5)
    assert torch.isnan(dW[:, width:]).all()
    _close(dW[:, :width], ref, 1e-5)
    dW2, _ = _wgrad(dense, sparse, F, D, g1, scale)
    assert torch.equal(dW2, dW[:, :width])
    dW3, _ = _wgrad(dense, sparse, F, D, g1, None)
    _close(dW3, 2 * ref, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [65536, 65521])
def test_top_wgrad_full_batch_on_the_gpu(B):
    """BASELINE's batch (and one that is no multiple of the 32-sample tile): 64 batch slices x 4 column groups, 32 tiles per
    workgroup; against the fp64 product over the materialised z, and twice the same bits."""
    _lib.use_native()
    dev = torch.device("cuda", 0)
    D, H, F = 16, 64, 26
    torch.manual_seed(B)
    dense, sparse, g1 = torch.randn(B, D, device=dev), torch.randn(B, F * D, device=dev), torch.randn(B, H, device=dev) / B
    ref = (g1.double().t() @ _torch_z(dense, sparse, D).double()).float()
    dW, width = _wgrad(dense, sparse, F, D, g1, None)
    _close(dW, ref, 1e-5)
    dW2, _ = _wgrad(dense, sparse, F, D, g1, None)
    assert torch.equal(dW, dW2)


def test_top_wgrad_without_dense_row_and_empty_batch(dev):
    D, H, F, B = 16, 64, 20, 45
    torch.manual_seed(3)
    sparse = torch.randn(B, F * D, device=dev)
    g1 = torch.randn(B, H, device=dev)
    X = sparse.view(B, F, D).double()
    iu = torch.triu_indices(F, F, offset=1)
    z = torch.cat([torch.bmm(X, X.transpose(1, 2))[:, iu[0], iu[1]], sparse.double()], dim=1)
    dW, width = _wgrad(None, sparse, F, D, g1, None)
    assert width == z.shape[1]
    _close(dW, (g1.double().t() @ z).float(), 1e-5)
    e = sparse[:0]
    dW0, _ = _wgrad(None, e, F, D, g1[:0], None)
    assert torch.count_nonzero(dW0) == 0


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_top_fwd_wave_orders_agree(dev, mode):
    """`it_fwd_stagger`: half of the waves build the next tile's row before the product / all before / all behind -- the same
    bits (the order only moves work between barriers), with and without the z store."""
    D, H, F, B = 16, 64, 26, 1100
    torch.manual_seed(5)
    L = _lib.lib()
    dense, sparse = torch.randn(B, D, device=dev), torch.randn(B, F * D, device=dev)
    width = 27 * 26 // 2 + 27 * D
    W1, b1 = torch.randn(H, width, device=dev) * 0.05, torch.randn(H, device=dev)

    def run(with_z):
        z = torch.empty(B, width, device=dev) if with_z else None
        y1 = torch.empty(B, H, device=dev)
        _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1,
                                                 _lib.ptr(z), width, _lib.ptr(y1), H, _lib.stream_ptr(sparse.device)), "fwd")
        return y1, z

    y_ref, z_ref = run(True)
    assert L.tzr_tune(b"it_fwd_stagger", mode) == 0
    try:
        y_a, z_a = run(True)
        y_b, _ = run(False)
    finally:
        L.tzr_tune(b"it_fwd_stagger", 0)
    assert torch.equal(y_a, y_ref) and torch.equal(z_a, z_ref) and torch.equal(y_b, y_ref)


def test_top_unsupported_shapes(dev):
    L = _lib.lib()
    assert L.tzr_dot_interaction_top_supported(26, 16, 1, 64) == 1
    assert L.tzr_dot_interaction_top_supported(26, 16, 1, 32) == 0
    assert L.tzr_dot_interaction_top_supported(26, 8, 1, 64) == 0
    assert L.tzr_dot_interaction_top_supported(31, 16, 1, 64) == 1  # n = 32: 31 pair blocks + 32 rows = 63 <= 64
    assert L.tzr_dot_interaction_top_supported(32, 16, 1, 64) == 0
    x = torch.zeros(4, 64, device=dev)
    rc = L.tzr_dot_interaction_top_fwd(None, 0, _lib.ptr(x), 64, 8, 8, 4, _lib.ptr(x), 64, None, 64, 1, None, 0, _lib.ptr(x), 64,
                                       _lib.stream_ptr(x.device))
    assert rc == -4  # TZR_ERR_UNSUPPORTED
    # more than 2^30 samples in one call: refused before anything is read (the kernels count samples in 32 bits)
    y = torch.zeros(4, 27 * 16, device=dev)
    rc = L.tzr_dot_interaction_top_fwd(_lib.ptr(y), 16, _lib.ptr(y), 26 * 16, 26, 16, (1 << 30) + 1, _lib.ptr(y), 783, None, 64, 1, None, 0,
                                       _lib.ptr(y), 64, _lib.stream_ptr(y.device))
    assert rc == -4
    rc = L.tzr_dot_interaction_top_bwd(_lib.ptr(y), 16, _lib.ptr(y), 26 * 16, 26, 16, (1 << 30) + 1, _lib.ptr(y), 64, 64, _lib.ptr(y), 783,
                                       None, _lib.ptr(y), 16, _lib.ptr(y), 26 * 16, _lib.stream_ptr(y.device))
    assert rc == -4


@pytest.mark.parametrize("B", [5, 100])
def test_interaction_top_loss_matches_unfused(dev, B):
    """DLRM head: the fused autograd function against top_loss(dot_interaction(...)), every gradient."""
    from torcheasyrec_amd.dense import interaction_top_fits, interaction_top_loss, top_loss
    from torcheasyrec_amd.interaction import dot_interaction

    D, F = 16, 26
    torch.manual_seed(B)
    width = 27 * 26 // 2 + 27 * D
    l1, l2, lo = torch.nn.Linear(width, 64).to(dev), torch.nn.Linear(64, 32).to(dev), torch.nn.Linear(32, 1).to(dev)
    dense = torch.randn(B, D, device=dev, requires_grad=True)
    sparse = torch.randn(B, F * D, device=dev, requires_grad=True)
    y = (torch.rand(B, device=dev) < 0.3).long()
    assert interaction_top_fits(dense, sparse, D, l1)
    ps = [dense, sparse, l1.weight, l1.bias, l2.weight, l2.bias, lo.weight, lo.bias]
    loss_ref, logits_ref = top_loss(dot_interaction(dense, sparse, D, True, True), l1, l2, lo, y)
    (loss_ref * 0.25).backward()
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    loss, logits = interaction_top_loss(dense, sparse, D, l1, l2, lo, y)
    torch.testing.assert_close(logits, logits_ref, rtol=1e-5, atol=1e-6)
    assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref)) + 1e-7
    (loss * 0.25).backward()
    for w, p in zip(want, ps):
        _close(p.grad, w, 1e-5)


@pytest.mark.parametrize("scaled", [False, True])
def test_interaction_top_loss_keeps_no_interaction_rows(dev, monkeypatch, scaled):
    """The fused DLRM head with the weight gradient of its first layer from tzr_dot_interaction_top_wgrad: the forward is
    called without a z buffer, and every gradient equals the variant that stores z for the GEMM library (`OWNED_WGRAD =
    False`) to 1e-5 of its largest entry -- for autograd's unit gradient and for a scaled loss."""
    from torcheasyrec_amd import dense as dn

    D, F, B = 16, 26, 70
    torch.manual_seed(21)
    width = 27 * 26 // 2 + 27 * D
    l1, l2, lo = torch.nn.Linear(width, 64).to(dev), torch.nn.Linear(64, 32).to(dev), torch.nn.Linear(32, 1).to(dev)
    x_d, x_s = torch.randn(B, D, device=dev, requires_grad=True), torch.randn(B, F * D, device=dev, requires_grad=True)
    y = (torch.rand(B, device=dev) < 0.3).long()
    ps = [x_d, x_s, l1.weight, l1.bias, l2.weight, l2.bias, lo.weight, lo.bias]
    allocated = []
    real_empty = torch.empty

    def spy_empty(*size, **kw):
        allocated.append(tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size))
        return real_empty(*size, **kw)

    grads, kept_z = {}, {}
    for owned in (True, False):
        monkeypatch.setattr(dn, "OWNED_WGRAD", owned)
        for p_ in ps:
            p_.grad = None
        allocated.clear()
        monkeypatch.setattr(torch, "empty", spy_empty)
        loss, _ = dn.interaction_top_loss(x_d, x_s, D, l1, l2, lo, y)
        monkeypatch.setattr(torch, "empty", real_empty)
        kept_z[owned] = (B, width) in allocated
        if scaled:
            (loss * 0.125).backward()
        else:
            with dn.root_loss():
                loss.backward(gradient=dn.unit_gradient(loss))
        grads[owned] = [p_.grad.clone() for p_ in ps]
    assert kept_z == {True: False, False: True}
    for a_, b_ in zip(grads[True], grads[False]):
        _close(a_, b_, 1e-5)


@pytest.mark.parametrize("fused_interaction", [True, False])
def test_root_loss_skips_the_unit_scaling_and_nothing_else(dev, fused_interaction, monkeypatch):
    """`dense.root_loss()`: when the fused loss itself is differentiated its incoming gradient is 1.0; the multi-tensor
    launch that multiplies the six parameter gradients by it is skipped -- on the autograd ENGINE's thread, where the
    backward of device tensors runs -- and every gradient stays bit-identical."""
    from torcheasyrec_amd import dense as dn
    from torcheasyrec_amd.interaction import dot_interaction

    D, F, B = 16, 26, 40
    torch.manual_seed(9)
    width = 27 * 26 // 2 + 27 * D
    l1, l2, lo = torch.nn.Linear(width, 64).to(dev), torch.nn.Linear(64, 32).to(dev), torch.nn.Linear(32, 1).to(dev)
    x_d, x_s = torch.randn(B, D, device=dev, requires_grad=True), torch.randn(B, F * D, device=dev, requires_grad=True)
    y = (torch.rand(B, device=dev) < 0.3).long()
    ps = [x_d, x_s, l1.weight, l1.bias, l2.weight, l2.bias, lo.weight, lo.bias]

    def loss_of():
        if fused_interaction:
            return dn.interaction_top_loss(x_d, x_s, D, l1, l2, lo, y)[0]
        return dn.top_loss(dot_interaction(x_d, x_s, D, True, True), l1, l2, lo, y)[0]

    calls = {"n": 0}
    real = torch._foreach_mul

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(torch, "_foreach_mul", counted)
    want = torch.autograd.grad(loss_of(), ps)
    assert calls["n"] == 1
    with dn.root_loss():
        l_ = loss_of()
        got = torch.autograd.grad(l_, ps, grad_outputs=dn.unit_gradient(l_))
    assert calls["n"] == 1  # not called again
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert not dn._loss_is_root()
    # a caller that announces a root loss but hands in ANOTHER gradient (a scaled loss: task weights, GradScaler, 1 / accumulation
    # steps) gets the scaled gradients: the unscaled path is taken only for the cached unit gradient itself
    with dn.root_loss():
        half = torch.autograd.grad(loss_of() * 0.5, ps)
        l_ = loss_of()
        other_one = torch.autograd.grad(l_, ps, grad_outputs=torch.ones_like(l_))
    assert calls["n"] == 3
    for a, b, c in zip(half, want, other_one):
        torch.testing.assert_close(a, 0.5 * b, rtol=1e-6, atol=1e-9)
        assert torch.equal(c, b)


def test_dlrm_predict_without_grad_uses_fused_first_layer(dev):
    """Inference (torch.no_grad) runs interaction + first top layer fused; logits equal the layer-wise path."""
    from torcheasyrec_amd.criteo import SPARSE_KEYS, criteo_tables, synthetic_batch
    from torcheasyrec_amd.dlrm import DLRM
    from torcheasyrec_amd.embedding import SparseOptimizerConfig

    rows = [50 + 3 * i for i in range(26)]
    torch.manual_seed(3)
    m = DLRM(criteo_tables(rows), SPARSE_KEYS, dense_dim=13, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.01))
    dense, kjt, _ = synthetic_batch(5, 48, rows)
    dense, kjt = dense.to(dev), kjt.to(dev)
    want = m(dense, kjt).detach()
    with torch.no_grad():
        got = m(dense, kjt)
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("B,wgs", [(600, 3), (97, 2), (256, 1), (33, 2), (16, 1), (40, 3), (70, 2)])
def test_top_persistent_loop_over_many_tiles(dev, B, wgs):
    """Few workgroups, many tiles each: the software pipeline (rows of tile t + G produced beside the product of tile t,
    double-buffered z tile, g1 / X fetched a tile ahead), a partial last tile, odd and even trip counts."""
    D, H, F = 16, 64, 26
    torch.manual_seed(B + wgs)
    L = _lib.lib()
    dense = torch.randn(B, D, device=dev, requires_grad=True)
    sparse = torch.randn(B, F * D, device=dev, requires_grad=True)
    width = 27 * 26 // 2 + D * 27
    lin = torch.nn.Linear(width, H).to(dev)
    g1 = torch.randn(B, H, device=dev)
    z_ref = _torch_z(dense, sparse, D)
    pre = lin(z_ref)
    (pre * g1).sum().backward()
    stream = _lib.stream_ptr(sparse.device)
    assert L.tzr_tune(b"it_wgs", wgs) == 0
    z = torch.full((B, width), float("nan"), device=dev)
    y1 = torch.empty(B, H, device=dev)
    _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(lin.weight), width,
                                             _lib.ptr(lin.bias), H, 0, _lib.ptr(z), width, _lib.ptr(y1), H, stream), "fwd")
    _close(y1, pre.detach(), 2e-6)
    _close(z, z_ref.detach(), 1e-6)
    # ... and the kernel that does not write z (the training step's: one barrier per tile, the sum of a tile's partials taken
    # a tile later out of the second set): the same sums in the same order, so the same bits
    for relu, order in ((0, 0), (1, 0), (1, 1), (0, 2), (1, 3)):  # (order of a wave's turn: it_fwd_stagger)
        y1z, y1n = torch.full((B, H), float("nan"), device=dev), torch.full((B, H), float("nan"), device=dev)
        assert L.tzr_tune(b"it_fwd_stagger", order) == 0
        try:
            for zz, out in ((z, y1z), (None, y1n)):
                _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(lin.weight), width,
                                                         _lib.ptr(lin.bias), H, relu, _lib.ptr(zz), width, _lib.ptr(out), H, stream), "fwd")
        finally:
            L.tzr_tune(b"it_fwd_stagger", 0)
        assert torch.equal(y1z, y1n)
        _close(y1n, (torch.relu(pre) if relu else pre).detach(), 2e-6)
    gd, gs = torch.full_like(dense, float("nan")), torch.full_like(sparse, float("nan"))
    _lib.check(L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H,
                                             _lib.ptr(lin.weight), width, None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, stream), "bwd")
    _close(gd, dense.grad, 1e-5)
    _close(gs, sparse.grad, 1e-5)


@pytest.mark.gpu
def test_top_kernels_at_full_batch_match_the_unfused_ops():
    """B = 65 536 (BASELINE config 1's global batch on one GPU), the DLRM-Criteo shape: the fused kernels against
    tzr_dot_interaction_fwd / _bwd + torch GEMMs on the same inputs -- z bit-identical (the same fmaf chains), y1 and
    the input gradients to 2e-6 / 1e-5 of the largest entry (another summation order of the 783-long products)."""
    _lib.use_native()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    B, D, F, H = 65536, 16, 26, 64
    torch.manual_seed(5)
    n = F + 1
    width = n * (n - 1) // 2 + D * n
    dense, sparse = torch.randn(B, D, device=dev), torch.randn(B, F * D, device=dev)
    W1, b1, g1 = torch.randn(H, width, device=dev) * 0.05, torch.randn(H, device=dev), torch.randn(B, H, device=dev)
    st = _lib.stream_ptr(dev)
    z_ref = torch.empty(B, width, device=dev)
    _lib.check(L.tzr_dot_interaction_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(z_ref), width, 1, 1, st), "fwd")
    y_ref = torch.relu(torch.addmm(b1, z_ref, W1.t()))
    dz = g1 @ W1
    gd_ref, gs_ref = torch.empty_like(dense), torch.empty_like(sparse)
    _lib.check(L.tzr_dot_interaction_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(dz), width, 1, 1, _lib.ptr(gd_ref), D,
                                         _lib.ptr(gs_ref), F * D, st), "bwd")
    z, y1 = torch.full((B, width), float("nan"), device=dev), torch.empty(B, H, device=dev)
    _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1,
                                             _lib.ptr(z), width, _lib.ptr(y1), H, st), "top_fwd")
    gd, gs = torch.full_like(dense, float("nan")), torch.full_like(sparse, float("nan"))
    _lib.check(L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width,
                                             None, _lib.ptr(gd), D, _lib.ptr(gs), F * D, st), "top_bwd")
    torch.cuda.synchronize()
    assert torch.equal(z, z_ref)
    _close(y1, y_ref, 2e-6)
    _close(gd, gd_ref, 1e-5)
    _close(gs, gs_ref, 1e-5)
    # deterministic: a second launch gives the same bits
    y2, gs2 = torch.empty_like(y1), torch.empty_like(gs)
    _lib.check(L.tzr_dot_interaction_top_fwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(W1), width, _lib.ptr(b1), H, 1,
                                             None, 0, _lib.ptr(y2), H, st), "top_fwd")
    _lib.check(L.tzr_dot_interaction_top_bwd(_lib.ptr(dense), D, _lib.ptr(sparse), F * D, F, D, B, _lib.ptr(g1), H, H, _lib.ptr(W1), width,
                                             None, _lib.ptr(gd), D, _lib.ptr(gs2), F * D, st), "top_bwd")
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(gs, gs2)


@pytest.mark.parametrize("F,with_dense", [(26, True), (27, False), (9, False)])
def test_top_kernels_with_row_strides_and_without_a_dense_row(dev, F, with_dense):
    """Inputs that are views into wider buffers (row strides larger than the row) and the no-dense-row form of the C
    entry points (X = the sparse rows only)."""
    D, H, B = 16, 64, 70
    torch.manual_seed(F)
    L = _lib.lib()
    n = F + (1 if with_dense else 0)
    width = n * (n - 1) // 2 + D * n
    dbuf = torch.randn(B, D + 8, device=dev)
    sbuf = torch.randn(B, F * D + 12, device=dev)
    gbuf = torch.randn(B, H + 4, device=dev)
    dense = dbuf[:, :D].detach().clone().requires_grad_(True) if with_dense else None
    sparse = sbuf[:, :F * D].detach().clone().requires_grad_(True)
    g1 = gbuf[:, :H]
    W1buf = torch.randn(H, width + 5, device=dev) * 0.1
    W1 = W1buf[:, :width]  # ldw > width
    b1 = torch.randn(H, device=dev)
    X = sparse.view(B, F, D) if not with_dense else torch.cat([dense.unsqueeze(1), sparse.view(B, F, D)], dim=1)
    iu = torch.triu_indices(n, n, offset=1, device=X.device)
    z_ref = torch.cat([torch.bmm(X, X.transpose(1, 2))[:, iu[0], iu[1]]] + ([dense] if with_dense else []) + [sparse], dim=1)
    pre = z_ref @ W1.t() + b1
    (pre * g1).sum().backward()
    st = _lib.stream_ptr(sparse.device)
    dptr, dstride = (_lib.ptr(dbuf), D + 8) if with_dense else (None, 0)
    zbuf = torch.full((B, width + 3), float("nan"), device=dev)
    ybuf = torch.empty(B, H + 2, device=dev)
    _lib.check(L.tzr_dot_interaction_top_fwd(dptr, dstride, _lib.ptr(sbuf), F * D + 12, F, D, B, _lib.ptr(W1buf), width + 5, _lib.ptr(b1), H,
                                             0, _lib.ptr(zbuf), width + 3, _lib.ptr(ybuf), H + 2, st), "fwd")
    with torch.no_grad():
        dbuf_ok = not with_dense or torch.equal(dbuf[:, :D], dense)
    assert dbuf_ok
    _close(ybuf[:, :H], pre.detach(), 2e-6)
    _close(zbuf[:, :width], z_ref.detach(), 1e-6)
    assert torch.isnan(zbuf[:, width:]).all()
    gdb = torch.full((B, D + 8), float("nan"), device=dev)
    gsb = torch.full((B, F * D + 12), float("nan"), device=dev)
    _lib.check(L.tzr_dot_interaction_top_bwd(dptr, dstride, _lib.ptr(sbuf), F * D + 12, F, D, B, _lib.ptr(gbuf), H + 4, H, _lib.ptr(W1buf),
                                             width + 5, None, _lib.ptr(gdb) if with_dense else None, D + 8, _lib.ptr(gsb), F * D + 12, st),
               "bwd")
    _close(gsb[:, :F * D], sparse.grad, 1e-5)
    assert torch.isnan(gsb[:, F * D:]).all()
    if with_dense:
        _close(gdb[:, :D], dense.grad, 1e-5)
        assert torch.isnan(gdb[:, D:]).all()


def test_every_feature_count_on_the_emulator(emu_path):
    """F = 1 .. 31 (every pair-block / row-block count the fused kernels dispatch on: the 48-block, 49-block and 64-block
    forms of the backward, the 12 + 1 and the generic K split of the forward) x B in {1, 19, 33}: emulator only, ~10 s."""
    _lib.use_library(emu_path)  # (the `dev` fixture of the other tests selects its library itself)
    dev = torch.device("cpu")
    for F in range(1, 32):
        for B in (1, 19, 33):
            test_top_fwd_bwd_match_torch(dev, B, F)
