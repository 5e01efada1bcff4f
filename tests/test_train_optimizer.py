"""TZRecOptimizer / dense gradient clipping (the optimizer seam): call pattern of the reference's train loop
(`optimizer.zero_grad(); loss.backward(); optimizer.step()`, /root/reference/tzrec/optim/optimizer.py:26-68,
/root/reference/tzrec/main.py:848-876) against plain torch on the same micro-batches."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd.config import load_pipeline_spec  # noqa: E402
from torcheasyrec_amd.optimizer import (GradClippingConfig, GradientClippingOptimizer, TZRecOptimizer,  # noqa: E402
                                        build_train_optimizer)


def _net(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 1))


def _batches(n):
    g = torch.Generator().manual_seed(3)
    return [(torch.randn(6, 5, generator=g), torch.randn(6, 1, generator=g)) for _ in range(n)]


@pytest.mark.parametrize("accum", [0, 1, 3])
def test_gradient_accumulation_steps_every_nth_call(accum):
    """zero_grad / step act on every accum-th call: the weights equal one plain step on the SUM of the
    micro-batch gradients (the reference does not rescale the loss)."""
    a, b = _net(), _net()
    opt = TZRecOptimizer(torch.optim.SGD(a.parameters(), lr=0.1), gradient_accumulation_steps=accum)
    ref = torch.optim.SGD(b.parameters(), lr=0.1)
    n = max(accum, 1)
    last = [p.detach().clone() for p in a.parameters()]
    for k, (x, y) in enumerate(_batches(2 * n)):
        opt.zero_grad()
        ((a(x) - y) ** 2).mean().backward()
        opt.step()
        if k % n == 0:
            ref.zero_grad()
        ((b(x) - y) ** 2).mean().backward()
        if (k + 1) % n == 0:
            ref.step()
            for p, q in zip(a.parameters(), b.parameters()):
                assert torch.equal(p, q)
            last = [p.detach().clone() for p in a.parameters()]
        else:  # between boundaries nothing moves
            assert all(torch.equal(p, q) for p, q in zip(a.parameters(), last))


@pytest.mark.parametrize("kind,norm_type", [("norm", 2.0), ("norm", float("inf")), ("value", 2.0), ("none", 2.0)])
def test_dense_gradient_clipping(kind, norm_type):
    a, b = _net(1), _net(1)
    opt = build_train_optimizer(torch.optim.SGD(a.parameters(), lr=0.5), GradClippingConfig(kind, 0.05, norm_type))
    ref = torch.optim.SGD(b.parameters(), lr=0.5)
    for x, y in _batches(3):
        opt.zero_grad()
        ((a(x) - y) ** 2).mean().backward()
        opt.step()
        ref.zero_grad()
        ((b(x) - y) ** 2).mean().backward()
        if kind == "norm":
            torch.nn.utils.clip_grad_norm_(list(b.parameters()), 0.05, norm_type=norm_type)
        elif kind == "value":
            torch.nn.utils.clip_grad_value_(list(b.parameters()), 0.05)
        ref.step()
    for p, q in zip(a.parameters(), b.parameters()):
        assert torch.equal(p, q)
    assert isinstance(opt, TZRecOptimizer) and isinstance(opt._optimizer, GradientClippingOptimizer) == (kind != "none")
    with pytest.raises(ValueError, match="Invalid clipping_type 'foo'"):
        build_train_optimizer(torch.optim.SGD(a.parameters(), lr=0.5), GradClippingConfig("foo"))


def test_grad_scaler_drives_the_step():
    class Scaler:
        def __init__(self):
            self.calls = []

        def step(self, opt):
            self.calls.append("step")
            opt.step()

        def update(self):
            self.calls.append("update")

    a = _net()
    sc = Scaler()
    opt = TZRecOptimizer(torch.optim.SGD(a.parameters(), lr=0.1), grad_scaler=sc, gradient_accumulation_steps=2)
    before = [p.detach().clone() for p in a.parameters()]
    for x, y in _batches(2):
        opt.zero_grad()
        ((a(x) - y) ** 2).mean().backward()
        opt.step()
    assert sc.calls == ["step", "update"] and not torch.equal(before[0], next(a.parameters()))
    assert opt.param_groups[0]["lr"] == 0.1  # schedulers reach the wrapped optimizer's groups


def test_fields_come_from_the_train_config():
    text = open(os.path.join(os.path.dirname(__file__), "golden", "deepfm_mini.config")).read()
    text = text.replace("train_config {", 'train_config {\n gradient_accumulation_steps: 4\n grad_clipping { clipping_type: "norm" max_gradient: 2.5 norm_type: inf }', 1)
    spec = load_pipeline_spec(text)
    assert spec.gradient_accumulation_steps == 4
    gc = spec.grad_clipping
    assert (gc.clipping_type, gc.max_gradient, gc.norm_type, gc.enable_global_grad_clip) == ("norm", 2.5, float("inf"), False)
    assert load_pipeline_spec(open(os.path.join(os.path.dirname(__file__), "golden", "deepfm_mini.config")).read()).grad_clipping is None
