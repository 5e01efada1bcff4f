"""The small dense layer stacks as whole-stack kernels (csrc/mlp_ops.hip) against torch autograd on the same
weights: the bottom MLP (tzr_mlp2_fwd / bwd) and the top MLP's tail through the loss (tzr_mlp_tail), then the
DLRM training step through `forward_loss` against the layer-by-layer path.  fp32: forward values to 1e-6, gradients
(sums over the batch in another order) to 1e-5 relative of the largest entry."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def _close(a, b, rtol=1e-5):
    scale = float(b.abs().max()) + 1e-12
    assert float((a - b).abs().max()) <= rtol * scale + 1e-9, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize("B,K0,H1,H2", [(1, 13, 64, 16), (63, 13, 64, 16), (200, 13, 64, 16), (1000, 32, 64, 32), (130, 3, 8, 4), (77, 5, 33, 7)])
def test_mlp2_matches_torch(dev, B, K0, H1, H2):
    from torcheasyrec_amd.dense import mlp2

    torch.manual_seed(B + K0)
    la, lb = torch.nn.Linear(K0, H1).to(dev), torch.nn.Linear(H1, H2).to(dev)
    x = torch.randn(B, K0, device=dev)
    g = torch.randn(B, H2, device=dev)
    ref = torch.relu(lb(torch.relu(la(x))))
    ref.backward(g)
    ps = (la.weight, la.bias, lb.weight, lb.bias)
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    out = mlp2(x, *ps)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    out.backward(g)
    for w, p in zip(want, ps):
        _close(p.grad, w)


@pytest.mark.parametrize("B,K,H1,H2,float_labels", [(1, 20, 64, 32, False), (333, 783, 64, 32, False), (64, 40, 64, 32, True), (129, 17, 24, 9, False)])
def test_top_loss_matches_torch(dev, B, K, H1, H2, float_labels):
    from torcheasyrec_amd.dense import top_loss

    torch.manual_seed(B)
    l1, l2, lo = torch.nn.Linear(K, H1).to(dev), torch.nn.Linear(H1, H2).to(dev), torch.nn.Linear(H2, 1).to(dev)
    z = torch.randn(B, K, device=dev, requires_grad=True)
    y = (torch.rand(B, device=dev) < 0.3)
    y = y.float() if float_labels else y.long()
    logits_ref = lo(torch.relu(l2(torch.relu(l1(z))))).squeeze(1)
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(logits_ref, y.float())
    (loss_ref * 0.25).backward()  # a gradient-accumulation style scale on the loss must reach every gradient
    ps = [z, l1.weight, l1.bias, l2.weight, l2.bias, lo.weight, lo.bias]
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    loss, logits = top_loss(z, l1, l2, lo, y)
    assert not logits.requires_grad
    torch.testing.assert_close(logits, logits_ref.detach(), rtol=1e-5, atol=1e-6)
    assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref)) + 1e-7
    (loss * 0.25).backward()
    for w, p in zip(want, ps):
        _close(p.grad, w)


def test_dlrm_forward_loss_matches_layerwise_step(dev):
    """One DLRM training step through forward_loss (fused bottom MLP + fused top tail + loss) and through
    forward + bce_with_logits with the fused stacks switched off: same loss, same dense gradients, same tables."""
    from torcheasyrec_amd import dlrm as dl
    from torcheasyrec_amd.criteo import NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch
    from torcheasyrec_amd.embedding import SparseOptimizerConfig

    rows = [min(r, 3000) for r in __import__("torcheasyrec_amd.criteo", fromlist=["CRITEO_ROWS"]).CRITEO_ROWS]
    B = 192
    dense, kjt, label = synthetic_batch(3, B, rows)
    res = []
    for fused in (True, False):
        dl._FUSED_MLP2, dl._FUSED_TOP_LOSS = fused, fused
        try:
            torch.manual_seed(11)
            m = dl.DLRM(criteo_tables(rows, init="seeded"), SPARSE_KEYS, NUM_DENSE, device=dev,
                        sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.05, initial_accumulator_value=0.1))
            if fused:
                loss, logits = m.forward_loss(dense.to(dev), kjt.to(dev), label.to(dev))
            else:
                logits = m(dense.to(dev), kjt.to(dev))
                loss = dl.bce_with_logits(logits, label.to(dev))
            loss.backward()
            res.append((float(loss), logits.detach().cpu(), [p.grad.detach().cpu().clone() for p in m.dense_parameters()],
                        {n: w.detach().cpu().clone() for n, w in m.ebc.table_weights().items()}))
        finally:
            dl._FUSED_MLP2, dl._FUSED_TOP_LOSS = True, True
    (la, ga, pa, wa), (lb, gb, pb, wb) = res
    assert abs(la - lb) <= 1e-6 * abs(lb) + 1e-7
    torch.testing.assert_close(ga, gb, rtol=1e-5, atol=1e-6)
    for a, b in zip(pa, pb):
        _close(a, b)
    for n in wa:
        torch.testing.assert_close(wa[n], wb[n], rtol=1e-5, atol=1e-7, msg=n)


@pytest.mark.parametrize("knob", [-1, 0, 2])
def test_top_loss_kernel_variants_agree(dev, knob):
    """The 64 -> 32 -> 1 tail on the matrix cores (mlp_mfma.hip; knob 2: two workgroups, several tiles per wave) and on
    the general LDS-tiled kernel (knob -1) against torch autograd."""
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.dense import top_loss

    B, K = 700, 48
    torch.manual_seed(11)
    l1, l2, lo = torch.nn.Linear(K, 64).to(dev), torch.nn.Linear(64, 32).to(dev), torch.nn.Linear(32, 1).to(dev)
    z = torch.randn(B, K, device=dev, requires_grad=True)
    y = (torch.rand(B, device=dev) < 0.3).long()
    logits_ref = lo(torch.relu(l2(torch.relu(l1(z))))).squeeze(1)
    loss_ref = torch.nn.functional.binary_cross_entropy_with_logits(logits_ref, y.float())
    loss_ref.backward()
    ps = [z, l1.weight, l1.bias, l2.weight, l2.bias, lo.weight, lo.bias]
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    assert _lib.lib().tzr_tune(b"mlp_mfma", knob) == 0
    loss, logits = top_loss(z, l1, l2, lo, y)
    torch.testing.assert_close(logits, logits_ref.detach(), rtol=1e-5, atol=1e-6)
    assert abs(float(loss) - float(loss_ref)) <= 2e-6 * abs(float(loss_ref)) + 1e-7
    loss.backward()
    for w, p in zip(want, ps):
        _close(p.grad, w)


@pytest.mark.parametrize("knob", [-1, 0, 2])
@pytest.mark.parametrize("B,K0", [(700, 13), (33, 16), (50, 5)])
def test_mlp2_kernel_variants_agree(dev, knob, B, K0):
    """The 13 -> 64 -> 16 stack on the matrix cores (knob 2: two workgroups, several tiles per wave) and on the general
    LDS-tiled kernels (knob -1) against torch autograd."""
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.dense import mlp2

    torch.manual_seed(B + K0)
    la, lb = torch.nn.Linear(K0, 64).to(dev), torch.nn.Linear(64, 16).to(dev)
    x = torch.randn(B, K0, device=dev)
    g = torch.randn(B, 16, device=dev)
    ref = torch.relu(lb(torch.relu(la(x))))
    ref.backward(g)
    ps = (la.weight, la.bias, lb.weight, lb.bias)
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = None
    assert _lib.lib().tzr_tune(b"mlp_mfma", knob) == 0
    out = mlp2(x, *ps)
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    out.backward(g)
    for w, p in zip(want, ps):
        _close(p.grad, w)


def test_every_tail_of_the_16_sample_tiles_on_the_emulator(emu_path):
    """The MFMA forms take 16 samples per wave and 4 waves per workgroup: B = 1 .. 35 and the sizes around 48, 64, 128
    (every partial tile / partial workgroup), bottom-stack inputs 1 .. 16 wide, both label types.  Emulator only, ~3 s."""
    from torcheasyrec_amd import _lib

    _lib.use_library(emu_path)  # (the `dev` fixture of the other tests selects its library itself)
    dev = torch.device("cpu")
    for B in list(range(1, 36)) + [47, 48, 49, 63, 64, 65, 127, 129]:
        for K0 in (1, 4, 13, 16):
            test_mlp2_matches_torch(dev, B, K0, 64, 16)
        for float_labels in (False, True):
            test_top_loss_matches_torch(dev, B, 40, 64, 32, float_labels)
