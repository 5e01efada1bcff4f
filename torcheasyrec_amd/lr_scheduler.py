"""Learning-rate schedules for the fused sparse optimizer and the dense optimizer.

Same class names, constructor arguments and step semantics as the reference
(/root/reference/tzrec/optim/lr_scheduler.py:26-272; built from the `learning_rate` oneof of a
`sparse_optimizer` / `dense_optimizer` block by tzrec/optim/optimizer_builder.py:154-176 and stepped
by the train loop, tzrec/main.py:542-544 per step and :597-599 per epoch).  The reference subclasses
torch's LRScheduler; the fused sparse optimizer here is not a torch Optimizer (the update happens
inside the backward kernels, which read the rate from a device scalar), so the base class restates
the part of LRScheduler the schedules rely on: the rate of step 0 is applied at construction, every
`step()` advances the counter and writes `param_groups[i]["lr"]`.  Works on anything with
`param_groups` (FusedSparseOptimizer, FusedDenseAdam, torch.optim optimizers).
"""
from __future__ import annotations

import bisect
import math
from typing import Dict, List, Sequence


class BaseLR:
    def __init__(self, optimizer, by_epoch: bool = False) -> None:
        self.optimizer = optimizer
        self._by_epoch = by_epoch
        for g in optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
        self.base_lrs: List[float] = [g["initial_lr"] for g in optimizer.param_groups]
        self._step_count = 0
        self._last_lr: List[float] = list(self.base_lrs)
        self.step()

    @property
    def by_epoch(self) -> bool:
        return self._by_epoch

    def _get_lr(self) -> List[float]:
        raise NotImplementedError

    def get_lr(self) -> List[float]:
        # a base rate of 0 freezes the group at 0 whatever the schedule says
        return [b if b == 0 else lr for b, lr in zip(self.base_lrs, self._get_lr())]

    def get_last_lr(self) -> List[float]:
        return self._last_lr

    def step(self) -> None:
        self._step_count += 1
        lrs = self.get_lr()
        for g, lr in zip(self.optimizer.param_groups, lrs):
            g["lr"] = lr
        self._last_lr = list(lrs)

    def state_dict(self) -> Dict[str, object]:
        return {"step_count": self._step_count, "base_lrs": list(self.base_lrs), "last_lr": list(self._last_lr)}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        self._step_count = int(sd["step_count"])
        self.base_lrs = list(sd["base_lrs"])
        self._last_lr = list(sd["last_lr"])
        for g, lr in zip(self.optimizer.param_groups, self._last_lr):
            g["lr"] = lr

    def _warmup(self, step: int, start: float, size: int) -> List[float]:
        scale = step / size
        return [(b - start) * scale + start for b in self.base_lrs]


class ConstantLR(BaseLR):
    def __init__(self, optimizer) -> None:
        super().__init__(optimizer, by_epoch=True)

    def _get_lr(self) -> List[float]:
        return self.base_lrs


class ExponentialDecayLR(BaseLR):
    def __init__(self, optimizer, decay_size: int, decay_factor: float, staircase: bool = True,
                 warmup_learning_rate: float = 0.0, warmup_size: int = 0, min_learning_rate: float = 0.0,
                 by_epoch: bool = False) -> None:
        self._decay_size, self._decay_factor, self._staircase = decay_size, decay_factor, staircase
        self._warmup_learning_rate, self._warmup_size, self._min_learning_rate = warmup_learning_rate, warmup_size, min_learning_rate
        super().__init__(optimizer, by_epoch=by_epoch)

    def _get_lr(self) -> List[float]:
        step = max(self._step_count - 1, 0)
        if step < self._warmup_size:
            return self._warmup(step, self._warmup_learning_rate, self._warmup_size)
        p = (step - self._warmup_size) / self._decay_size
        if self._staircase:
            p = math.floor(p)
        scale = math.pow(self._decay_factor, p)
        return [max(b * scale, self._min_learning_rate) for b in self.base_lrs]


class ManualStepLR(BaseLR):
    def __init__(self, optimizer, schedule_sizes: Sequence[int], learning_rates: Sequence[float], warmup: bool = False,
                 by_epoch: bool = False) -> None:
        self._schedule_sizes, self._learning_rates, self._warmup_flag = list(schedule_sizes), list(learning_rates), warmup
        super().__init__(optimizer, by_epoch=by_epoch)

    def _get_lr(self) -> List[float]:
        step = max(self._step_count - 1, 0)
        idx = bisect.bisect_left(self._schedule_sizes, step)
        if idx > 0:
            return [self._learning_rates[idx - 1] for _ in self.base_lrs]
        if self._warmup_flag:
            scale = step / self._schedule_sizes[0]
            return [(self._learning_rates[0] - b) * scale + b for b in self.base_lrs]
        return self.base_lrs


class CosineAnnealingLR(BaseLR):
    def __init__(self, optimizer, T_max: int, min_learning_rate: float = 0.0, warmup_learning_rate: float = 0.0,
                 warmup_size: int = 0, by_epoch: bool = False) -> None:
        if T_max <= 0:
            raise ValueError(f"T_max must be positive, got {T_max}")
        self._T_max, self._min_learning_rate = T_max, min_learning_rate
        self._warmup_learning_rate, self._warmup_size = warmup_learning_rate, warmup_size
        super().__init__(optimizer, by_epoch=by_epoch)

    def _get_lr(self) -> List[float]:
        step = max(self._step_count - 1, 0)
        if step < self._warmup_size:
            return self._warmup(step, self._warmup_learning_rate, self._warmup_size)
        t = min(step - self._warmup_size, self._T_max)
        c = 0.5 * (1 + math.cos(math.pi * t / self._T_max))
        return [self._min_learning_rate + (b - self._min_learning_rate) * c for b in self.base_lrs]


class CosineAnnealingWarmRestartsLR(BaseLR):
    def __init__(self, optimizer, T_0: int, T_mult: int = 1, min_learning_rate: float = 0.0,
                 warmup_learning_rate: float = 0.0, warmup_size: int = 0, by_epoch: bool = False) -> None:
        if T_0 <= 0:
            raise ValueError(f"T_0 must be positive, got {T_0}")
        self._T_0, self._T_mult, self._min_learning_rate = T_0, T_mult, min_learning_rate
        self._warmup_learning_rate, self._warmup_size = warmup_learning_rate, warmup_size
        super().__init__(optimizer, by_epoch=by_epoch)

    def _get_lr(self) -> List[float]:
        step = max(self._step_count - 1, 0)
        if step < self._warmup_size:
            return self._warmup(step, self._warmup_learning_rate, self._warmup_size)
        elapsed = step - self._warmup_size
        if self._T_mult == 1:
            t_cur, t_i = elapsed % self._T_0, self._T_0
        else:
            n = math.floor(math.log(elapsed / self._T_0 * (self._T_mult - 1) + 1, self._T_mult))
            mult_n = self._T_mult ** n
            t_i = self._T_0 * mult_n
            t_cur = elapsed - self._T_0 * (mult_n - 1) // (self._T_mult - 1)
        c = 0.5 * (1 + math.cos(math.pi * t_cur / t_i))
        return [self._min_learning_rate + (b - self._min_learning_rate) * c for b in self.base_lrs]


_BLOCKS = {
    "constant_learning_rate": (ConstantLR, {}),
    "exponential_decay_learning_rate": (ExponentialDecayLR, {
        "decay_size": (int, None), "decay_factor": (float, 0.95), "staircase": (bool, True), "warmup_learning_rate": (float, 0.0),
        "warmup_size": (int, 0), "min_learning_rate": (float, 0.0), "by_epoch": (bool, False)}),
    "manual_step_learning_rate": (ManualStepLR, {
        "schedule_sizes": ([int], None), "learning_rates": ([float], None), "warmup": (bool, False), "by_epoch": (bool, False)}),
    "cosine_annealing_learning_rate": (CosineAnnealingLR, {
        "T_max": (int, None), "min_learning_rate": (float, 0.0), "warmup_learning_rate": (float, 0.0), "warmup_size": (int, 0),
        "by_epoch": (bool, False)}),
    "cosine_annealing_warm_restarts_learning_rate": (CosineAnnealingWarmRestartsLR, {
        "T_0": (int, None), "T_mult": (int, 1), "min_learning_rate": (float, 0.0), "warmup_learning_rate": (float, 0.0),
        "warmup_size": (int, 0), "by_epoch": (bool, False)}),
}


def _as(kind, v):
    if kind is bool:
        return v if isinstance(v, bool) else str(v).lower() == "true"
    if kind is float:  # proto `float` fields reach the reference's schedules rounded to fp32
        import numpy as np

        return float(np.float32(v))
    return kind(v)


def create_scheduler(optimizer, optimizer_block) -> BaseLR:
    """The `learning_rate` oneof of a parsed `sparse_optimizer {...}` / `dense_optimizer {...}` block
    (config.Msg) -> schedule; field defaults from protos/optimizer.proto:211-268.  A block without
    the oneof gets ConstantLR."""
    for name, (cls, fields) in _BLOCKS.items():
        if optimizer_block is not None and optimizer_block.has(name):
            msg = optimizer_block.one(name)
            kw = {}
            for f, (kind, default) in fields.items():
                if isinstance(kind, list):
                    kw[f] = [_as(kind[0], x) for x in msg.many(f)]
                elif msg.has(f):
                    kw[f] = _as(kind, msg.one(f))
                elif default is not None:
                    kw[f] = default
                else:
                    raise ValueError(f"{name}: field {f} is required")
            return cls(optimizer, **kw)
    return ConstantLR(optimizer)
