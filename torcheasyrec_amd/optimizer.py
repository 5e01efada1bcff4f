"""Optimizer wrappers of the train loop (the optimizer seam).

`TZRecOptimizer` mirrors /root/reference/tzrec/optim/optimizer.py:26-68: gradient accumulation and an
optional grad scaler around the DENSE optimizer -- the sparse update already happened inside
`loss.backward()` (fused in the backward kernels, every micro-step, as fbgemm's in-backward optimizer
does), so `step()` here only ever touches dense parameters.  `GradientClippingOptimizer` is the dense
gradient clipping the reference wraps around it from `train_config.grad_clipping`
(/root/reference/tzrec/main.py:851-868, torchrec.optim.clipping [upstream 1.7.0]: "norm" / "value" /
"none").  Host-side control flow; the arithmetic is torch's clip_grad_* on the dense gradients.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Iterable, List, Optional

import torch


class OptimizerWrapper:
    """Forwards everything to the wrapped optimizer (torchrec.optim.OptimizerWrapper's role)."""

    def __init__(self, optimizer) -> None:
        self._optimizer = optimizer

    @property
    def param_groups(self):
        return self._optimizer.param_groups

    def zero_grad(self, set_to_none: bool = False) -> None:
        self._optimizer.zero_grad(set_to_none=set_to_none)

    def step(self, closure: Any = None) -> None:
        self._optimizer.step(closure) if closure is not None else self._optimizer.step()

    def state_dict(self):
        return self._optimizer.state_dict()

    def load_state_dict(self, sd) -> None:
        self._optimizer.load_state_dict(sd)


def _params_of(optimizer) -> List[torch.Tensor]:
    return [p for g in optimizer.param_groups for p in g["params"]]


class GradientClippingOptimizer(OptimizerWrapper):
    """Clip the dense gradients right before the wrapped step.  `clipping`: "norm" (total norm of all
    gradients scaled down to `max_gradient`, `norm_type` 2.0 or inf), "value" (every element clamped to
    +-max_gradient) or "none".  `enable_global_grad_clip`: the norm is taken over the ranks of
    `process_group` too -- for data-parallel dense parameters whose gradients were already averaged
    every rank holds the same values and the local norm IS the global one, so nothing is exchanged."""

    def __init__(self, optimizer, clipping: str = "none", max_gradient: float = 1.0, norm_type: float = 2.0,
                 enable_global_grad_clip: bool = False) -> None:
        super().__init__(optimizer)
        clipping = str(clipping).lower()
        if clipping not in ("norm", "value", "none"):
            raise ValueError(f"Invalid clipping_type '{clipping}'. Valid values are: norm, value, none")
        self._clipping, self._max_gradient, self._norm_type = clipping, float(max_gradient), float(norm_type)
        self._global = bool(enable_global_grad_clip)

    def step(self, closure: Any = None) -> None:
        from .dense import materialize_pending

        materialize_pending([p.grad for p in _params_of(self._optimizer) if p.grad is not None])  # (clipping reads finished gradients)
        params = [p for p in _params_of(self._optimizer) if p.grad is not None]
        if params and self._clipping == "norm":
            torch.nn.utils.clip_grad_norm_(params, self._max_gradient, norm_type=self._norm_type)
        elif params and self._clipping == "value":
            torch.nn.utils.clip_grad_value_(params, self._max_gradient)
        super().step(closure)


class TZRecOptimizer(OptimizerWrapper):
    """Gradient accumulation / grad scaler around the dense optimizer (reference :26-68, same arguments):
    `zero_grad` and `step` act on every `gradient_accumulation_steps`-th call only."""

    def __init__(self, optimizer, grad_scaler: Optional[Any] = None, gradient_accumulation_steps: int = 0) -> None:
        super().__init__(optimizer)
        self._step = 0
        self._grad_scaler = grad_scaler
        self._gradient_accumulation_steps = int(gradient_accumulation_steps)

    def _boundary(self) -> bool:
        return self._gradient_accumulation_steps <= 1 or self._step % self._gradient_accumulation_steps == 0

    def zero_grad(self, set_to_none: bool = False) -> None:
        if self._boundary():
            self._optimizer.zero_grad(set_to_none=set_to_none)

    def step(self, closure: Any = None) -> None:
        self._step += 1
        if self._boundary():
            if self._grad_scaler is not None:
                self._grad_scaler.step(self._optimizer)
                self._grad_scaler.update()
            else:
                super().step(closure)


@dataclass
class GradClippingConfig:
    """train.proto:21-30"""

    clipping_type: str = "none"
    max_gradient: float = 1.0
    norm_type: float = 2.0
    enable_global_grad_clip: bool = False


def grad_clipping_from_msg(msg) -> GradClippingConfig:
    nt = msg.one("norm_type", 2.0)
    return GradClippingConfig(str(msg.one("clipping_type", "none")), float(msg.one("max_gradient", 1.0)),
                              float("inf") if str(nt).lower() in ("inf", "infinity") else float(nt),
                              bool(msg.one("enable_global_grad_clip", False)))


def build_train_optimizer(dense_optimizer, grad_clipping: Optional[GradClippingConfig] = None, gradient_accumulation_steps: int = 0,
                          grad_scaler: Optional[Any] = None) -> TZRecOptimizer:
    """The wrapping order of tzrec/main.py:848-876: clipping (if any) inside, TZRecOptimizer outside."""
    opt = dense_optimizer
    if grad_clipping is not None and grad_clipping.clipping_type.lower() != "none":  # an unknown type raises ValueError
        opt = GradientClippingOptimizer(opt, grad_clipping.clipping_type, grad_clipping.max_gradient, grad_clipping.norm_type,
                                        grad_clipping.enable_global_grad_clip)
    return TZRecOptimizer(opt, grad_scaler=grad_scaler, gradient_accumulation_steps=gradient_accumulation_steps)
