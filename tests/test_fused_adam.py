"""tzr_dense_adam_fused (csrc/adam_fused.hip): the dense optimizer that takes gradients as they lie -- finished tensors, rows of
partial sums (tzr_mlp2_bwd_parts), the slices of the first top-MLP layer's weight gradient (tzr_dot_interaction_top_wgrad_parts).
The additions are the finishing launches' own, so a training run with `FusedDenseAdam(fuse_finish=True)` must equal the run
with separate finishing launches BIT FOR BIT; gradients that nobody steps are written out as tensors all the same."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd import dense  # noqa: E402
from torcheasyrec_amd.criteo import NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.dense import FusedDenseAdam  # noqa: E402
from torcheasyrec_amd.dlrm import DLRM  # noqa: E402
from torcheasyrec_amd.embedding import SparseOptimizerConfig  # noqa: E402


@pytest.fixture(autouse=True)
def _reset_flag():
    dense.FUSE_FINISH = False
    dense._PENDING.clear()
    yield
    dense.FUSE_FINISH = False
    dense._PENDING.clear()


def _train(dev, fuse, steps, B=96, read_grads_at=None):
    torch.manual_seed(0)
    rows = [min(r, 300) for r in [40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209, 11938, 155, 4, 976, 14,
                                  40000000, 40000000, 40000000, 590152, 12973, 108, 36]]
    model = DLRM(criteo_tables(rows, init="seeded"), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="sgd", lr=0.01))
    params = list(model.dense_parameters())
    dense.FUSE_FINISH = False
    opt = FusedDenseAdam(params, lr=1e-2, weight_decay=1e-3, fuse_finish=fuse)
    assert dense.FUSE_FINISH == fuse
    grads = None
    for s in range(steps):
        d, kjt, y = synthetic_batch(s, B, rows)
        loss, _ = model.forward_loss(d.to(dev), kjt.to(dev), y.to(dev))
        with dense.root_loss():
            loss.backward(gradient=dense.unit_gradient(loss))
        if read_grads_at == s:
            dense.materialize_pending()  # (a reader of finished gradients in front of the optimizer)
            assert not dense._PENDING
            grads = [p.grad.detach().cpu().clone() for p in params]
        opt.step()
        assert not dense._PENDING
        opt.zero_grad(set_to_none=True)
    return [p.detach().cpu().clone() for p in params], opt._state.cpu().clone(), grads, float(loss.detach())


def test_training_with_gradients_left_as_partial_sums_is_bit_identical(dev):
    pa, sa, _, la = _train(dev, False, 3)
    pb, sb, _, lb = _train(dev, True, 3)
    assert la == lb
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)
    assert torch.equal(sa, sb) and bool((sa[:, 0] == 3).all()) and bool((sa[:, 1] == 0).all())  # steps counted, arrival counters back at zero


def test_pending_gradients_can_be_written_out_before_the_step(dev):
    _, _, ga, _ = _train(dev, False, 2, read_grads_at=1)
    pb, _, gb, _ = _train(dev, True, 2, read_grads_at=1)
    pa, _, _, _ = _train(dev, False, 2)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    for a, b in zip(pa, pb):  # ... and the step behind it takes the written-out tensors: the same parameters again
        assert torch.equal(a, b)


def test_more_tensors_than_one_launch_takes_and_tensors_without_a_gradient(dev):
    """40 tensors (two launches of <= 32), every third without a gradient: its step count does not move (torch counts steps per
    parameter), the others follow torch.optim.Adam"""
    torch.manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(int(n), device=dev)) for n in np.random.default_rng(0).integers(1, 3000, size=40)]
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = FusedDenseAdam(ps, lr=3e-3)
    topt = torch.optim.Adam(ref, lr=3e-3)
    for step in range(3):
        for i, (p, r) in enumerate(zip(ps, ref)):
            if i % 3 == step % 3:
                p.grad = r.grad = None
                continue
            g = torch.randn(p.shape, generator=torch.Generator().manual_seed(100 * step + i))
            p.grad, r.grad = g.to(dev), g.clone()
        opt.step()
        topt.step()
    for i, (p, r) in enumerate(zip(ps, ref)):
        torch.testing.assert_close(p.detach().cpu(), r.detach(), rtol=2e-6, atol=2e-7)
    assert opt._state[:, 0].cpu().tolist() == [2.0] * 40 and bool((opt._state[:, 1] == 0).all())


def test_a_gradient_that_would_be_accumulated_is_finished_as_a_tensor(dev):
    """parameters that still hold a gradient when the backward runs (no zero_grad: autograd ADDS the new gradient to the old one)
    do not get theirs as partial sums: two backward passes + one step == the same with the finishing launches"""
    def run(fuse):
        torch.manual_seed(0)
        rows = [200] * 26
        model = DLRM(criteo_tables(rows, init="seeded"), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="sgd", lr=0.01))
        dense.FUSE_FINISH = False
        opt = FusedDenseAdam(list(model.dense_parameters()), lr=1e-2, fuse_finish=fuse)
        for s in range(2):
            d, kjt, y = synthetic_batch(s, 64, rows)
            loss, _ = model.forward_loss(d.to(dev), kjt.to(dev), y.to(dev))
            with dense.root_loss():
                loss.backward(gradient=dense.unit_gradient(loss))  # (no zero_grad between the two)
        opt.step()
        assert not dense._PENDING
        return [p.detach().cpu().clone() for p in model.dense_parameters()]

    for a, b in zip(run(False), run(True)):
        assert torch.equal(a, b)


def test_only_parameters_of_a_live_fusing_optimizer_are_deferred(dev):
    """dense.FUSE_FINISH is process-wide and stays up; the promise behind it is ONE optimizer's.  A model stepped by another
    optimizer (here: torch's Adam) in the same process must get finished gradients -- its backward must not leave partial sums
    nobody will add up."""
    import gc

    from torcheasyrec_amd import dense
    from torcheasyrec_amd.dense import mlp2

    torch.manual_seed(0)
    keep = FusedDenseAdam([torch.nn.Parameter(torch.zeros(4, device=dev))], lr=1e-2, fuse_finish=True)  # raises the flag
    assert dense.FUSE_FINISH
    Wa, ba = torch.nn.Parameter(torch.randn(64, 13, device=dev) * 0.1), torch.nn.Parameter(torch.zeros(64, device=dev))
    Wb, bb = torch.nn.Parameter(torch.randn(16, 64, device=dev) * 0.1), torch.nn.Parameter(torch.zeros(16, device=dev))
    x = torch.randn(96, 13, device=dev)
    n0 = len(dense._PENDING)
    mlp2(x, Wa, ba, Wb, bb).sum().backward()
    assert len(dense._PENDING) == n0  # nothing deferred: these parameters belong to no fusing optimizer
    ref = torch.relu(torch.relu(x.cpu() @ Wa.detach().cpu().t()) @ Wb.detach().cpu().t())
    gWb = torch.autograd.grad  # (finished tensors: compare one of them with torch)
    xr, War, Wbr = x.cpu(), Wa.detach().cpu().requires_grad_(True), Wb.detach().cpu().requires_grad_(True)
    torch.relu(torch.relu(xr @ War.t()) @ Wbr.t()).sum().backward()
    torch.testing.assert_close(Wb.grad.cpu(), Wbr.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(Wa.grad.cpu(), War.grad, rtol=1e-5, atol=1e-6)
    # ... and once the fusing optimizer is gone its parameters are not deferred either
    own = [torch.nn.Parameter(t.detach().clone()) for t in (Wa, ba, Wb, bb)]
    o = FusedDenseAdam(own, lr=1e-2, fuse_finish=True)
    mlp2(x, *own).sum().backward()
    assert len(dense._PENDING) > n0
    o.step()
    for q in own:
        q.grad = None
    del o
    gc.collect()
    n1 = len(dense._PENDING)
    mlp2(x, *own).sum().backward()
    assert len(dense._PENDING) == n1
    del keep


def test_pack_gradients_equals_concatenation_and_fails_loudly_on_a_lost_partial(dev):
    """dense.pack_gradients: finished tensors and partial sums into one flat buffer in one launch == torch.cat of the finished
    gradients, bit for bit; a gradient that was left as partial sums but is not among the tensors handed in (autograd passed it on
    as a copy) is an error, never a silently unwritten slice."""
    from torcheasyrec_amd.dense import mlp2, pack_gradients

    torch.manual_seed(0)
    x = torch.randn(96, 13, device=dev)

    def grads(fuse):
        own = [torch.nn.Parameter(t) for t in (torch.randn(64, 13, device=dev) * 0.1, torch.zeros(64, device=dev),
                                               torch.randn(16, 64, device=dev) * 0.1, torch.zeros(16, device=dev))]
        extra = torch.nn.Parameter(torch.randn(7, 5, device=dev))
        o = FusedDenseAdam(own + [extra], lr=1e-2, fuse_finish=fuse)
        loss = mlp2(x, *own).sum() + (extra * extra).sum()
        gs = torch.autograd.grad(loss, own + [extra])
        return o, gs

    torch.manual_seed(1)
    o1, g_plain = grads(False)
    want = torch.cat([g.reshape(-1) for g in g_plain])
    torch.manual_seed(1)
    o2, g_fused = grads(True)
    assert len(dense._PENDING) >= 4
    flat = pack_gradients(list(g_fused))
    assert flat is not None and torch.equal(flat.cpu(), want.cpu())
    assert not any(e[3] == dense._GENERATION[0] for e in dense._PENDING.values())
    torch.manual_seed(1)
    o3, g_lost = grads(True)
    with pytest.raises(RuntimeError, match="partial sums"):
        pack_gradients([g.clone() if i == 2 else g for i, g in enumerate(g_lost)])  # (tensor 2 arrives as a copy)
    dense._PENDING.clear()
    del o1, o2, o3


def test_linear_relu_bias_gradients_go_into_the_optimizers_launch(dev):
    """dlrm.MLP (Linear + bias + ReLU layers) under FusedDenseAdam(fuse_finish=True): the bias gradients stay the mask kernel's
    partial rows (tzr_relu_bwd_colsum_parts) and are added up inside tzr_dense_adam_fused -- the same training run as with the
    finishing launches, to the rounding of a different summation order (<= 1e-6 relative on the parameters after 5 steps)."""
    from torcheasyrec_amd import dlrm
    from torcheasyrec_amd.dlrm import MLP

    if dev.type == "cpu":
        pytest.skip("the Linear + ReLU autograd function is the GPU path of dlrm.MLP")

    def run(fuse):
        torch.manual_seed(0)
        mlp = MLP(96, [256, 128, 64]).to(dev)
        dense.FUSE_FINISH = False
        opt = FusedDenseAdam(list(mlp.parameters()), lr=1e-2, fuse_finish=fuse)
        deferred = 0
        for s in range(5):
            x = torch.randn(4096, 96, generator=torch.Generator().manual_seed(s)).to(dev)
            n0 = len(dense._PENDING)
            mlp(x).square().mean().backward()
            deferred += len(dense._PENDING) - n0
            opt.step()
            opt.zero_grad()
        assert not dense._PENDING
        return [p.detach().cpu().clone() for p in mlp.parameters()], deferred

    (pa, da), (pb, db) = run(False), run(True)
    assert da == 0 and db == 5 * 3  # three layers' bias gradients per step
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)


def test_a_shared_layer_under_fuse_finish_fails_loudly(dev):
    """the same parameters twice in one backward pass: autograd would add two gradients of which the first is unwritten partial
    sums -- refused with an error naming the remedy, not trained on garbage"""
    from torcheasyrec_amd.dense import mlp2

    torch.manual_seed(0)
    own = [torch.nn.Parameter(t) for t in (torch.randn(64, 13, device=dev) * 0.1, torch.zeros(64, device=dev),
                                           torch.randn(16, 64, device=dev) * 0.1, torch.zeros(16, device=dev))]
    o = FusedDenseAdam(own, lr=1e-2, fuse_finish=True)
    x1, x2 = torch.randn(96, 13, device=dev), torch.randn(96, 13, device=dev)
    with pytest.raises(RuntimeError, match="shared layer"):
        (mlp2(x1, *own).sum() + mlp2(x2, *own).sum()).backward()
    dense._PENDING.clear()
    dense._DEFERRED.clear()
    del o


def test_an_abandoned_backward_pass_does_not_look_like_a_shared_layer(dev):
    """gradients dropped by hand (`p.grad = None`, no step, no zero_grad) and a new backward pass: the parameters' entries belong
    to another pass -- deferred again, trained normally, no 'shared layer' error"""
    from torcheasyrec_amd.dense import mlp2

    torch.manual_seed(0)
    own = [torch.nn.Parameter(t) for t in (torch.randn(64, 13, device=dev) * 0.1, torch.zeros(64, device=dev),
                                           torch.randn(16, 64, device=dev) * 0.1, torch.zeros(16, device=dev))]
    o = FusedDenseAdam(own, lr=1e-2, fuse_finish=True)
    x = torch.randn(96, 13, device=dev)
    mlp2(x, *own).sum().backward()
    for p in own:
        p.grad = None  # the pass is abandoned
    before = [p.detach().clone() for p in own]
    mlp2(x, *own).sum().backward()
    o.step()
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, own))
    assert bool(all(torch.isfinite(p).all() for p in own))
    dense._PENDING.clear()
    del o
