"""Dense glue of a training step behind the C ABI: fused BCE-with-logits (loss + gradient in one
two launches) and a two-launch Adam over all dense parameters.

At DLRM-Criteo's batch 65536 PyTorch runs BCEWithLogitsLoss forward+backward as ~12 and fused Adam as
~4 launches of 3-5 us each (profiles/r01e/kernel_stats.csv); under a captured hipGraph that is pure
launch serialisation.  Same math as the torch ops they replace
(/root/reference/tzrec/models/rank_model.py:190-191,233-240; /root/reference/tzrec/optim/optimizer.py:56-68):
tests compare against torch to 1e-6.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import _lib


class _BceLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, sample_weight):
        x = logits.contiguous().float()
        y = labels.contiguous()
        if y.dtype not in (torch.float32, torch.int32, torch.int64):
            y = y.float()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        grad = torch.empty_like(x)
        w = None if sample_weight is None else sample_weight.contiguous().float()
        L = _lib.lib()
        ws = _lib.workspace(L.tzr_bce_logits_workspace(x.numel()), x.device)
        _lib.check(L.tzr_bce_logits(_lib.ptr(x), _lib.ptr(y), y.element_size(), 1 if y.is_floating_point() else 0,
                                    _lib.ptr(w), x.numel(), _lib.ptr(loss), _lib.ptr(grad), _lib.ptr(ws), ws.numel(),
                                    _lib.stream_ptr(x.device)), "tzr_bce_logits")
        ctx.save_for_backward(grad)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (grad,) = ctx.saved_tensors
        return (grad * grad_out).view(ctx.shape), None, None


def bce_with_logits(logits: torch.Tensor, labels: torch.Tensor, sample_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """BCEWithLogitsLoss(reduction="mean"); with `sample_weight`: mean(loss_i * w_i)."""
    return _BceLogitsFn.apply(logits, labels, sample_weight)


def _rows16(t: torch.Tensor) -> torch.Tensor:
    """Row-major view the float4 kernels can read: unit column stride, 16-byte aligned rows (a
    gradient slice of a torch.cat is neither)."""
    if t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    return t.clone(memory_format=torch.contiguous_format)  # .contiguous() keeps odd strides of 1-row tensors


def relu_bwd_colsum(grad_y: torch.Tensor, y: torch.Tensor):
    """(grad_y * (y > 0), its column sums): ReLU backward + bias gradient of a Linear+ReLU layer."""
    B, N = y.shape
    gy, y = _rows16(grad_y), _rows16(y)
    g = torch.empty(B, N, dtype=torch.float32, device=y.device)
    col = torch.empty(N, dtype=torch.float32, device=y.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_relu_bwd_colsum_workspace(B, N), y.device)
    _lib.check(L.tzr_relu_bwd_colsum(_lib.ptr(gy), gy.stride(0), _lib.ptr(y), y.stride(0), B, N, _lib.ptr(g), g.stride(0),
                                     _lib.ptr(col), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(y.device)), "tzr_relu_bwd_colsum")
    return g, col


def head_bwd(grad_y: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, need_grad_x: bool = True):
    """Backward of Linear(in, 1): (grad_x [B, in] | None, grad_weight [1, in], grad_bias [1])."""
    B, N = x.shape
    gy = grad_y.reshape(B)
    xs = _rows16(x)
    gx = torch.empty(B, N, dtype=torch.float32, device=x.device) if need_grad_x else None
    wb = torch.empty(N + 4, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_head_bwd_workspace(B, N), x.device)
    _lib.check(L.tzr_head_bwd(_lib.ptr(gy), gy.stride(0), _lib.ptr(xs), xs.stride(0), _lib.ptr(weight.reshape(-1)), B, N,
                              _lib.ptr(gx), 0 if gx is None else gx.stride(0), _lib.ptr(wb), _lib.ptr(ws), ws.numel(),
                              _lib.stream_ptr(x.device)), "tzr_head_bwd")
    return gx, wb[:N].unsqueeze(0), wb[N:N + 1]


class FusedDenseAdam:
    """torch.optim.Adam (amsgrad off) for the dense parameters, two launches per step regardless of the
    number of tensors.  `param_groups[0]["lr"]` may be changed between steps (it is mirrored into a
    device scalar, so a captured hipGraph sees the new value)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
                raise ValueError("FusedDenseAdam needs contiguous float32 parameters on one device")
        self.param_groups = [{"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": float(weight_decay),
                              "params": self.params}]
        self.device = dev
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self._state = torch.zeros(len(self.params), 3, dtype=torch.float32, device=dev)  # per tensor: step, 1-b1^t, 1-b2^t
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self._lr_host = float(lr)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def sync_lr(self) -> None:
        """Mirror ``param_groups[0]["lr"]`` into the device scalar the kernel reads.  `step` does it itself outside a
        graph capture; a captured step must not (the fill would be replayed): call this before every replay."""
        g = self.param_groups[0]
        if g["lr"] != self._lr_host:
            self._lr_dev.fill_(g["lr"])
            self._lr_host = g["lr"]

    def step(self, grads: Optional[List[torch.Tensor]] = None) -> None:
        g = self.param_groups[0]
        if not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            self.sync_lr()
        rows = []
        for i, p in enumerate(self.params):
            gr = p.grad if grads is None else grads[i]
            if gr is None:
                continue
            if gr.dtype != torch.float32 or not gr.is_contiguous():
                gr = gr.contiguous().float()
            rows.append((p, gr, self.exp_avg[i], self.exp_avg_sq[i], self._state[i]))
        if not rows:
            return
        tab = (_lib.TzrAdamTensor * len(rows))()
        for i, (p, gr, m, v, st) in enumerate(rows):
            tab[i].param, tab[i].grad, tab[i].exp_avg, tab[i].exp_avg_sq = _lib.ptr(p.data), _lib.ptr(gr), _lib.ptr(m), _lib.ptr(v)
            tab[i].state, tab[i].numel = _lib.ptr(st), p.numel()
        b1, b2 = g["betas"]
        _lib.check(_lib.lib().tzr_dense_adam(tab, len(rows), _lib.ptr(self._lr_dev), g["lr"], b1, b2,
                                             g["eps"], g["weight_decay"], _lib.stream_ptr(self.device)), "tzr_dense_adam")

    def state_dict(self) -> dict:
        return {"state": {i: {"step": self._state[i, 0].clone(), "exp_avg": m, "exp_avg_sq": v}
                          for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))},
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd: dict) -> None:
        b1, b2 = self.param_groups[0]["betas"]
        for i, st in sd["state"].items():
            self.exp_avg[int(i)].copy_(st["exp_avg"])
            self.exp_avg_sq[int(i)].copy_(st["exp_avg_sq"])
            t = float(st["step"])
            self._state[int(i)] = torch.tensor([t, 1.0 - b1 ** t, 1.0 - b2 ** t])
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v


def sync_learning_rates(model, optimizer=None) -> None:
    """Before replaying a hipGraph that contains update kernels: copy every learning rate a scheduler may have changed
    (fused sparse optimizers of the model's collections, the dense optimizer behind its wrappers) into the device
    scalars the captured kernels read.  No-ops when nothing changed; never call it under capture."""
    seen = set()
    mods = list(model.modules()) if hasattr(model, "modules") else []
    for m in mods:
        for name in ("fused_optimizer",):
            try:
                fo = getattr(m, name, None)
            except Exception:
                fo = None
            if fo is not None and id(fo) not in seen and hasattr(fo, "sync_lr"):
                seen.add(id(fo))
                fo.sync_lr()
    opt = optimizer
    hops = 0
    while opt is not None and hops < 8:
        if hasattr(opt, "sync_lr"):
            opt.sync_lr()
            break
        opt = getattr(opt, "_optimizer", None)
        hops += 1
