"""Fused BCE-with-logits and the two-launch Adam against the torch ops they replace (fp32, 1e-6)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd.dense import FusedDenseAdam, bce_with_logits  # noqa: E402


@pytest.mark.parametrize("B,label_dtype,weighted", [(1, torch.float32, False), (1500, torch.int64, False),
                                                    (2049, torch.float32, True), (1100, torch.int32, False)])
def test_bce_matches_torch(dev, B, label_dtype, weighted):
    g = torch.Generator().manual_seed(B)
    x = (torch.randn(B, generator=g) * 6).requires_grad_(True)  # includes saturated logits
    x.data[0] = 40.0
    x.data[-1] = -40.0
    y = (torch.rand(B, generator=g) < 0.3).to(label_dtype)
    w = torch.rand(B, generator=g) + 0.5 if weighted else None
    ref = torch.nn.functional.binary_cross_entropy_with_logits(x.double(), y.double(), weight=None if w is None else w.double())
    ref.backward()
    gref, x.grad = x.grad.clone(), None
    xd = x.detach().to(dev).requires_grad_(True)
    loss = bce_with_logits(xd, y.to(dev), None if w is None else w.to(dev))
    (loss * 3.0).backward()  # upstream gradient is honoured
    torch.testing.assert_close(loss.detach().cpu().double(), ref.detach(), rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(xd.grad.cpu().double(), 3.0 * gref.double(), rtol=2e-6, atol=1e-9)
    assert loss.shape == ()


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_follows_torch_adam(dev, wd):
    torch.manual_seed(0)
    shapes = [(13, 64), (64,), (64, 16), (16,), (300, 7), (1,)] + [(3, 5)] * 28  # > 32 tensors: two launch pairs
    ref = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-2, weight_decay=wd)
    o = FusedDenseAdam(mine, lr=1e-2, weight_decay=wd)
    for step in range(7):
        if step == 4:  # a scheduler changes the learning rate
            o_ref.param_groups[0]["lr"] = 3e-3
            o.param_groups[0]["lr"] = 3e-3
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(step * 100 + p.numel()))
            p.grad = gr.clone()
            q.grad = gr.to(dev)
        if step == 5:
            ref[2].grad = None  # a parameter without gradient is skipped
            mine[2].grad = None
        o_ref.step()
        o.step()
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-6, atol=2e-7)
    sd = o.state_dict()
    o2 = FusedDenseAdam([torch.nn.Parameter(q.detach().clone()) for q in mine], lr=1.0)
    o2.load_state_dict(sd)
    assert o2.param_groups[0]["lr"] == 3e-3 and float(o2._state[0, 0]) == 7.0 and float(o2._state[2, 0]) == 6.0
    assert torch.equal(o2.exp_avg[4], o.exp_avg[4])


@pytest.mark.gpu
def test_mlp_with_relu_in_the_gemm_epilogue_matches_linear_then_relu():
    from torcheasyrec_amd import dlrm

    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    mlp = dlrm.MLP(37, [64, 16]).to(dev)
    x = torch.randn(513, 37, device=dev, requires_grad=True)
    g = torch.randn(513, 16, device=dev)
    out = {}
    for fused in (True, False):
        dlrm._FUSED_RELU = fused
        for p in mlp.parameters():
            p.grad = None
        x.grad = None
        y = mlp(x)
        (y * g).sum().backward()
        out[fused] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mlp.parameters()]
    dlrm._FUSED_RELU = True
    for a, b in zip(out[True], out[False]):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_output_layer_as_gemv_matches_linear():
    from torcheasyrec_amd import dlrm

    dev = torch.device("cuda", 0)
    torch.manual_seed(1)
    lin = dlrm.OutputLinear(32, 1).to(dev)
    x = torch.randn(1000, 32, device=dev, requires_grad=True)
    g = torch.randn(1000, device=dev)
    out = {}
    for fast in (True, False):
        dlrm._GEMV_OUTPUT = fast
        lin.weight.grad = lin.bias.grad = x.grad = None
        y = lin(x).squeeze(1)
        (y * g).sum().backward()
        out[fast] = [y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()]
    dlrm._GEMV_OUTPUT = True
    for a, b in zip(out[True], out[False]):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)  # 1000-term fp32 sums in a different order


@pytest.mark.parametrize("B,N", [(1, 4), (37, 16), (1000, 64), (5000, 32), (300, 1024), (70000, 8)])
def test_relu_bwd_colsum_matches_torch(dev, B, N):
    from torcheasyrec_amd.dense import relu_bwd_colsum

    g = torch.Generator().manual_seed(B + N)
    y = torch.relu(torch.randn(B, N, generator=g))
    gy_full = torch.randn(B, N + 5, generator=g)
    gy = gy_full[:, 1:N + 1] if B % 2 else gy_full[:, :N]  # row-strided; odd B: misaligned rows as well
    ref = gy * (y > 0)
    got, col = relu_bwd_colsum(gy.to(dev) if dev.type != "cpu" else gy, y.to(dev))
    assert torch.equal(got.cpu(), ref)
    torch.testing.assert_close(col.cpu().double(), ref.double().sum(0), rtol=1e-5, atol=1e-5 * (B ** 0.5))


@pytest.mark.parametrize("B,N", [(1, 4), (999, 32), (3000, 64), (70000, 8)])
def test_head_bwd_matches_torch(dev, B, N):
    from torcheasyrec_amd.dense import head_bwd

    g = torch.Generator().manual_seed(B * 7 + N)
    x = torch.randn(B, N, generator=g)
    w = torch.randn(1, N, generator=g)
    gy = torch.randn(B, 1, generator=g) / B
    gx, gw, gb = head_bwd(gy.to(dev), x.to(dev), w.to(dev))
    assert gx.shape == (B, N) and gw.shape == (1, N) and gb.shape == (1,)
    torch.testing.assert_close(gx.cpu(), gy @ w, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(gw.cpu().double(), gy.double().t() @ x.double(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gb.cpu().double(), gy.double().sum(0), rtol=1e-4, atol=1e-7)
    gx2, _, _ = head_bwd(gy.to(dev), x.to(dev), w.to(dev), need_grad_x=False)
    assert gx2 is None


@pytest.mark.parametrize("B,N", [(1, 4), (999, 32), (3000, 64), (70000, 8)])
def test_head_bwd_relu_matches_torch(dev, B, N):
    """the score layer's backward chained with the mask + bias gradient of the ReLU layer below it (tzr_head_bwd_relu)"""
    from torcheasyrec_amd.dense import head_bwd_relu

    g = torch.Generator().manual_seed(B * 5 + N)
    x = torch.relu(torch.randn(B, N, generator=g))
    w = torch.randn(1, N, generator=g)
    gy = torch.randn(B, 1, generator=g) / B
    got, gw, gb, col = head_bwd_relu(gy.to(dev), x.to(dev), w.to(dev))
    ref = (gy @ w) * (x > 0)
    assert got.shape == (B, N) and gw.shape == (1, N) and gb.shape == (1,) and col.shape == (N,)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(gw.cpu().double(), gy.double().t() @ x.double(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(gb.cpu().double(), gy.double().sum(0), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(col.cpu().double(), ref.double().sum(0), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("N,K,H,cap", [(1, 64, 256, 0), (16, 16, 64, 0), (37, 32, 128, 0), (1000, 64, 256, 0), (1000, 64, 256, 3),
                                        (515, 64, 64, 2), (2049, 16, 256, 5), (300, 32, 64, 1), (70000, 64, 128, 0)])
def test_linear_bwd_relu_matches_torch(dev, N, K, H, cap):
    """(g_in W) masked by the ReLU below + its column sums in one launch (tzr_linear_bwd_relu, exact-fp32 MFMA) against the three
    torch ops it replaces; `cap` workgroups: many tiles per workgroup (the double-buffered tile loop) on small inputs"""
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.dense import linear_bwd_relu, linear_bwd_relu_supported

    g = torch.Generator().manual_seed(N + K + H)
    y = torch.relu(torch.randn(N, H, generator=g))
    W = torch.randn(K, H, generator=g) / K ** 0.5
    gi_full = torch.randn(N, K + 8, generator=g)
    gi = gi_full[:, 4:K + 4] if N % 2 else gi_full[:, :K]  # row-strided input
    assert linear_bwd_relu_supported(gi.to(dev), W.to(dev))
    assert not linear_bwd_relu_supported(gi.to(dev), torch.randn(K, 96).to(dev))
    _lib.check(_lib.lib().tzr_tune(b"linear_bwd_wg", cap), "tzr_tune")
    try:
        got, col = linear_bwd_relu(gi.to(dev), W.to(dev), y.to(dev))
        again, col2 = linear_bwd_relu(gi.to(dev), W.to(dev), y.to(dev))
    finally:
        _lib.check(_lib.lib().tzr_tune(b"linear_bwd_wg", 0), "tzr_tune")
    ref = (gi.double() @ W.double()) * (y > 0)
    scale = float(ref.abs().max()) + 1e-30
    assert float((got.cpu().double() - ref).abs().max()) <= 2e-6 * scale  # fp32 products, 64-term sums
    assert torch.equal((got != 0).cpu() | (ref == 0), torch.ones(N, H, dtype=torch.bool))  # the mask itself is exact
    assert torch.equal(got.cpu()[y == 0], torch.zeros(int((y == 0).sum())))
    torch.testing.assert_close(col.cpu().double(), ref.sum(0), rtol=1e-4, atol=2e-6 * scale * N ** 0.5)
    assert torch.equal(got, again) and torch.equal(col, col2)  # fixed summation order


@pytest.mark.parametrize("B,K,n", [(1, 4, 1), (37, 64, 1), (1000, 64, 3), (999, 200, 2), (513, 256, 5), (300, 512, 8), (70, 1024, 3),
                                   (20000, 16, 4), (8192, 128, 3)])
def test_skinny_linear_matches_torch(dev, B, K, n):
    """Linear layers with <= 8 output units (logits, MMoE gates): tzr_skinny_linear_fwd / _bwd against nn.functional.linear and its
    autograd, through the module that the rank models use (dlrm.OutputLinear)"""
    from torcheasyrec_amd import dlrm
    from torcheasyrec_amd.dense import skinny_linear_bwd, skinny_linear_fwd, skinny_linear_ok

    g = torch.Generator().manual_seed(B + K + n)
    x_full = torch.randn(B, K + 8, generator=g)
    x = (x_full[:, 4:K + 4] if B % 2 else x_full[:, :K]).to(dev)  # row-strided input
    lin = dlrm.OutputLinear(K, n).to(dev)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(n, K, generator=g) / K ** 0.5)
        lin.bias.copy_(torch.randn(n, generator=g))
    assert skinny_linear_ok(x, lin.weight)
    gy = torch.randn(B, n, generator=g).to(dev) / B
    xr = x.detach().cpu().double().requires_grad_(True)
    wr, br = lin.weight.detach().cpu().double().requires_grad_(True), lin.bias.detach().cpu().double().requires_grad_(True)
    ref = torch.nn.functional.linear(xr, wr, br)
    ref.backward(gy.cpu().double())
    xd = x.detach().clone().requires_grad_(True)
    y = lin(xd)
    assert y.shape == (B, n) and y.grad_fn is not None and "Skinny" in type(y.grad_fn).__name__
    y.backward(gy)
    scale = float(ref.detach().abs().max())
    assert float((y.detach().cpu().double() - ref.detach()).abs().max()) <= 2e-6 * scale + 1e-7
    torch.testing.assert_close(xd.grad.cpu().double(), xr.grad, rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(lin.weight.grad.cpu().double(), wr.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(lin.bias.grad.cpu().double(), br.grad, rtol=1e-4, atol=1e-7)
    # the entry points themselves: no bias, no input gradient
    y2 = skinny_linear_fwd(x, lin.weight.detach(), None)
    torch.testing.assert_close(y2.cpu().double(), (ref - br).detach(), rtol=1e-5, atol=2e-6 * scale + 1e-7)
    gx2, gw2, gb2 = skinny_linear_bwd(gy, x, lin.weight.detach(), need_grad_x=False)
    assert gx2 is None and torch.equal(gw2, lin.weight.grad) and torch.equal(gb2, lin.bias.grad)  # deterministic


@pytest.mark.parametrize("B,H,E,T", [(1, 4, 1, 1), (37, 8, 3, 2), (1000, 128, 3, 2), (513, 264, 8, 4), (70, 1024, 2, 1), (4099, 64, 5, 3)])
def test_moe_mix_matches_the_stacked_form(dev, B, H, E, T):
    """softmax + mixing of all tasks of an MMoE in one launch per direction (tzr_moe_mix_fwd / _bwd) against the reference's
    literal form: stack, softmax, batched matmul (tzrec/modules/mmoe.py:63-76) and its autograd"""
    from torcheasyrec_amd.dense import moe_mix, moe_mix_ok

    g = torch.Generator().manual_seed(B + H + E + T)
    xs = [torch.randn(B, H, generator=g) for _ in range(E)]
    ls = [torch.randn(B, E, generator=g) * 2 for _ in range(T)]
    gos = [torch.randn(B, H, generator=g) for _ in range(T)]
    xr = [x.double().requires_grad_(True) for x in xs]
    lr = [l.double().requires_grad_(True) for l in ls]
    st = torch.stack(xr, dim=1)
    refs = [torch.matmul(torch.softmax(l, dim=1).unsqueeze(1), st).squeeze(1) for l in lr]
    torch.autograd.backward(refs, [go.double() for go in gos])
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    ld = [l.to(dev).requires_grad_(True) for l in ls]
    assert moe_mix_ok(ld, xd)
    outs = moe_mix(ld, xd)
    torch.autograd.backward(outs, [go.to(dev) for go in gos])
    for t in range(T):
        torch.testing.assert_close(outs[t].detach().cpu().double(), refs[t].detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(ld[t].grad.cpu().double(), lr[t].grad, rtol=1e-4, atol=1e-5 * H ** 0.5)
    for e in range(E):
        torch.testing.assert_close(xd[e].grad.cpu().double(), xr[e].grad, rtol=1e-5, atol=1e-6)
    assert not moe_mix_ok(ld, [x[:, :H - 1] for x in xd]) if H > 4 else True


def test_mlp_proto_fields_build_the_reference_perceptron():
    """use_bn / use_ln / dropout_ratio / activation / bias of the MLP proto (tzrec/protos/module.proto:4-17)
    reach the module (config-built towers used to read hidden_units only): same layer sequence as
    tzrec/modules/mlp.py Perceptron -- Linear (no bias under use_bn) -> BN | LN -> activation -> Dropout."""
    import torch
    from torch import nn

    from torcheasyrec_amd.config import parse_text_proto
    from torcheasyrec_amd.dlrm import MLP
    from torcheasyrec_amd.rank_model import mlp_from_msg

    m = mlp_from_msg(10, parse_text_proto('hidden_units: [8, 4] use_bn: true activation: "nn.GELU" dropout_ratio: [0.5, 0.25]'))
    kinds = [type(x) for x in m.mlp]
    assert kinds == [nn.Linear, nn.BatchNorm1d, nn.GELU, nn.Dropout, nn.Linear, nn.BatchNorm1d, nn.GELU, nn.Dropout]
    assert m.mlp[0].bias is None and m.mlp[3].p == 0.5 and m.mlp[7].p == 0.25
    m = mlp_from_msg(10, parse_text_proto("hidden_units: [8] use_ln: true bias: false"))
    assert [type(x) for x in m.mlp] == [nn.Linear, nn.LayerNorm, nn.ReLU] and m.mlp[0].bias is None
    m = mlp_from_msg(10, parse_text_proto("hidden_units: [8, 4]"))
    assert m._plain and [type(x) for x in m.mlp] == [nn.Linear, nn.ReLU, nn.Linear, nn.ReLU]
    # same numbers as the reference layer stack with shared parameters
    torch.manual_seed(0)
    ours = MLP(6, [5, 3], use_ln=True, activation="nn.Tanh")
    ref = nn.Sequential(nn.Linear(6, 5), nn.LayerNorm(5), nn.Tanh(), nn.Linear(5, 3), nn.LayerNorm(3), nn.Tanh())
    ref.load_state_dict({k.replace("mlp.", ""): v for k, v in ours.state_dict().items()})
    x = torch.randn(7, 6)
    assert torch.equal(ours(x), ref(x))
    try:
        MLP(4, [2], use_bn=True, use_ln=True)
        raise AssertionError("use_bn + use_ln accepted")
    except ValueError:
        pass
    try:
        MLP(4, [2], activation="Dice")
        raise AssertionError("unknown activation accepted")
    except NotImplementedError:
        pass


def test_unit_gradient_and_weight_gradient_policy():
    """`dense.unit_gradient`: one cached 1.0 per (device, dtype, shape) for the root of the backward pass;
    `dense._owned_wgrad`: the own weight-gradient kernel unless its 32-bit sample offsets would not fit."""
    import torch

    from torcheasyrec_amd import dense as dn

    loss = torch.tensor(0.25)
    one = dn.unit_gradient(loss)
    assert one is dn.unit_gradient(torch.tensor(3.0)) and float(one) == 1.0 and one.shape == loss.shape
    x = torch.tensor([2.0], requires_grad=True)
    (x * x).sum().backward(gradient=dn.unit_gradient((x * x).sum()))
    assert float(x.grad) == 4.0
    assert dn._owned_wgrad(65536, 416) and dn._owned_wgrad(1 << 20, 416)
    assert not dn._owned_wgrad(1 << 24, 416)  # 2^24 x 416 floats: offsets past 2^32
    keep = dn.OWNED_WGRAD
    try:
        dn.OWNED_WGRAD = False
        assert not dn._owned_wgrad(8192, 416)
    finally:
        dn.OWNED_WGRAD = keep
