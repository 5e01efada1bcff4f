"""Dense glue of a training step behind the C ABI: fused BCE-with-logits (loss + gradient in one
two launches) and a two-launch Adam over all dense parameters.

At DLRM-Criteo's batch 65536 PyTorch runs BCEWithLogitsLoss forward+backward as ~12 and fused Adam as
~4 launches of 3-5 us each (profiles/r01e/kernel_stats.csv); under a captured hipGraph that is pure
launch serialisation.  Same math as the torch ops they replace
(/root/reference/tzrec/models/rank_model.py:190-191,233-240; /root/reference/tzrec/optim/optimizer.py:56-68):
tests compare against torch to 1e-6.
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Iterable, List, Optional

import ctypes as C

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


def relu_bwd_colsum(grad_y: torch.Tensor, y: torch.Tensor, defer_for=None):
    """(grad_y * (y > 0), its column sums): ReLU backward + bias gradient of a Linear+ReLU layer.  `defer_for` = (the bias
    parameter,): with a live FusedDenseAdam(fuse_finish=True) stepping it the column sums are left as the kernel's per-workgroup
    partial rows for the optimizer's own launch (no finishing launch; the returned tensor is unwritten: see FUSE_FINISH)."""
    B, N = y.shape
    gy, y = _rows16(grad_y), _rows16(y)
    g = torch.empty(B, N, dtype=torch.float32, device=y.device)
    col = torch.empty(N, dtype=torch.float32, device=y.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_relu_bwd_colsum_workspace(B, N), y.device)
    if defer_for is not None and _defer_finish(y.device, defer_for):
        G = C.c_int(0)
        _lib.check(L.tzr_relu_bwd_colsum_parts(_lib.ptr(gy), gy.stride(0), _lib.ptr(y), y.stride(0), B, N, _lib.ptr(g), g.stride(0),
                                               _lib.ptr(ws), ws.numel(), C.byref(G), _lib.stream_ptr(y.device)), "tzr_relu_bwd_colsum_parts")
        _PENDING[col.data_ptr()] = ("rows", (ws,), (G.value, N, 0), _GENERATION[0], _owner_id(defer_for), tuple(id(q) for q in defer_for))
        return g, col
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


def head_bwd_relu(grad_y: torch.Tensor, x: torch.Tensor, weight: torch.Tensor):
    """Backward of Linear(in, 1) on the output x of a ReLU layer, chained with that layer's mask and bias gradient
    (tzr_head_bwd_relu): (g = grad_y * w * (x > 0) [B, in], grad_weight [1, in], grad_bias [1], column sums of g [in])."""
    B, N = x.shape
    gy = grad_y.reshape(B)
    xs = _rows16(x)
    g = torch.empty(B, N, dtype=torch.float32, device=x.device)
    sums = torch.empty(2 * N + 4, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_head_bwd_relu_workspace(B, N), x.device)
    _lib.check(L.tzr_head_bwd_relu(_lib.ptr(gy), gy.stride(0), _lib.ptr(xs), xs.stride(0), _lib.ptr(weight.reshape(-1)), B, N,
                                   _lib.ptr(g), g.stride(0), _lib.ptr(sums), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)),
               "tzr_head_bwd_relu")
    return g, sums[:N].unsqueeze(0), sums[N:N + 1], sums[N + 4:]


class _MoeMixFn(torch.autograd.Function):
    """out_t = softmax(logits_t) . experts for every task of a multi-gate mixture of experts in one launch per direction
    (tzr_moe_mix_fwd / _bwd, csrc/moe_ops.hip); inputs: T gate-logit tensors [B, E], then E expert outputs [B, H]."""

    @staticmethod
    def forward(ctx, n_tasks, *tensors):
        logits = [t.contiguous() for t in tensors[:n_tasks]]
        experts = [_rows16(t) for t in tensors[n_tasks:]]
        B, H = experts[0].shape
        E, dev = len(experts), experts[0].device
        m = _lib.TzrMoeMix()
        m.B, m.H, m.n_experts, m.n_tasks = B, H, E, n_tasks
        outs = [torch.empty(B, H, dtype=torch.float32, device=dev) for _ in range(n_tasks)]
        probs = [torch.empty(B, E, dtype=torch.float32, device=dev) for _ in range(n_tasks)]
        for e, x in enumerate(experts):
            m.expert[e], m.expert_stride[e] = _lib.ptr(x), x.stride(0)
        for t in range(n_tasks):
            m.logits[t], m.logits_stride[t] = _lib.ptr(logits[t]), logits[t].stride(0)
            m.probs[t] = _lib.ptr(probs[t])
            m.out[t], m.out_stride[t] = _lib.ptr(outs[t]), outs[t].stride(0)
        _lib.check(_lib.lib().tzr_moe_mix_fwd(_lib.C.byref(m), _lib.stream_ptr(dev)), "tzr_moe_mix_fwd")
        ctx.save_for_backward(*probs, *experts)
        ctx.n_tasks = n_tasks
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        T = ctx.n_tasks
        probs, experts = ctx.saved_tensors[:T], ctx.saved_tensors[T:]
        B, H = experts[0].shape
        E, dev = len(experts), experts[0].device
        gs = [torch.zeros(B, H, dtype=torch.float32, device=dev) if g is None else _rows16(g.float()) for g in gouts]
        m = _lib.TzrMoeMix()
        m.B, m.H, m.n_experts, m.n_tasks = B, H, E, T
        dx = [torch.empty(B, H, dtype=torch.float32, device=dev) for _ in range(E)]
        dl = [torch.empty(B, E, dtype=torch.float32, device=dev) for _ in range(T)]
        for e, x in enumerate(experts):
            m.expert[e], m.expert_stride[e] = _lib.ptr(x), x.stride(0)
            m.d_expert[e], m.d_expert_stride[e] = _lib.ptr(dx[e]), dx[e].stride(0)
        for t in range(T):
            m.probs[t] = _lib.ptr(probs[t])
            m.grad_out[t], m.grad_out_stride[t] = _lib.ptr(gs[t]), gs[t].stride(0)
            m.d_logits[t], m.d_logits_stride[t] = _lib.ptr(dl[t]), dl[t].stride(0)
        _lib.check(_lib.lib().tzr_moe_mix_bwd(_lib.C.byref(m), _lib.stream_ptr(dev)), "tzr_moe_mix_bwd")
        return (None, *dl, *dx)


def moe_mix_ok(logits, experts) -> bool:
    x = experts[0]
    return bool(0 < len(experts) <= _lib.MOE_MAX_EXPERTS and 0 < len(logits) <= _lib.MOE_MAX_TASKS and x.dim() == 2 and x.shape[0] > 0
                and x.shape[1] % 4 == 0 and x.shape[1] <= 4096 and all(t.dtype == torch.float32 and t.shape == x.shape for t in experts)
                and all(t.dtype == torch.float32 and t.shape == (x.shape[0], len(experts)) for t in logits))


def moe_mix(logits, experts):
    """[softmax(logits_t) . experts for t] -- the mixing step of MMoE (tzrec/modules/mmoe.py:63-76) -- one launch for all tasks."""
    return list(_MoeMixFn.apply(len(logits), *logits, *experts))


SKINNY_MAX_OUT = 8  # output units up to which a Linear layer takes tzr_skinny_linear_* (csrc/dense_ops.hip)


def skinny_linear_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    n, K = weight.shape
    return bool(x.dim() == 2 and x.shape[0] > 0 and x.dtype == torch.float32 and weight.dtype == torch.float32 and n <= SKINNY_MAX_OUT
                and K % 4 == 0 and K <= 1024 and weight.stride(1) == 1 and weight.stride(0) % 4 == 0)


def skinny_linear_fwd(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """x W^T + b for a Linear layer with <= 8 output units in one pass over x (tzr_skinny_linear_fwd)."""
    B, K = x.shape
    n = weight.shape[0]
    xs = _rows16(x)
    y = torch.empty(B, n, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tzr_skinny_linear_fwd(_lib.ptr(xs), xs.stride(0), _lib.ptr(weight), weight.stride(0), _lib.ptr(bias), B, K, n,
                                                _lib.ptr(y), y.stride(0), _lib.stream_ptr(x.device)), "tzr_skinny_linear_fwd")
    return y


def skinny_linear_bwd(grad_y: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, need_grad_x: bool = True, defer_for=None):
    """(grad_x [B, K] | None, grad_weight [n, K], grad_bias [n]) of a Linear layer with <= 8 output units (tzr_skinny_linear_bwd).
    `defer_for` = the layer's parameters: stepped by a live FusedDenseAdam(fuse_finish=True), their gradients stay the kernel's
    partial rows for the optimizer's launch (the returned tensors are unwritten: see FUSE_FINISH)."""
    B, K = x.shape
    n = weight.shape[0]
    gy = grad_y if grad_y.stride(1) == 1 or n == 1 else grad_y.contiguous()
    xs = _rows16(x)
    gx = torch.empty(B, K, dtype=torch.float32, device=x.device) if need_grad_x else None
    wb = torch.empty(n * K + (n + 3) // 4 * 4, dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_skinny_linear_bwd_workspace(B, K, n), x.device)
    if defer_for is not None and _defer_finish(x.device, defer_for):
        G, P = C.c_int(0), C.c_int(0)
        _lib.check(L.tzr_skinny_linear_bwd_parts(_lib.ptr(gy), gy.stride(0), _lib.ptr(xs), xs.stride(0), _lib.ptr(weight), weight.stride(0), B, K,
                                                 n, _lib.ptr(gx), 0 if gx is None else gx.stride(0), _lib.ptr(ws), ws.numel(), C.byref(G),
                                                 C.byref(P), _lib.stream_ptr(x.device)), "tzr_skinny_linear_bwd_parts")
        gw, gb = wb[:n * K].view(n, K), wb[n * K:n * K + n]
        _PENDING[gw.data_ptr()] = ("rows", (ws, wb), (G.value, P.value, 0), _GENERATION[0], _owner_id(defer_for), tuple(id(q) for q in defer_for))
        _PENDING[gb.data_ptr()] = ("rows", (ws, wb), (G.value, P.value, n * K), _GENERATION[0], _owner_id(defer_for), tuple(id(q) for q in defer_for))
        return gx, gw, gb
    _lib.check(L.tzr_skinny_linear_bwd(_lib.ptr(gy), gy.stride(0), _lib.ptr(xs), xs.stride(0), _lib.ptr(weight), weight.stride(0), B, K, n,
                                       _lib.ptr(gx), 0 if gx is None else gx.stride(0), _lib.ptr(wb), _lib.ptr(ws), ws.numel(),
                                       _lib.stream_ptr(x.device)), "tzr_skinny_linear_bwd")
    return gx, wb[:n * K].view(n, K), wb[n * K:n * K + n]


def linear_bwd_relu_supported(g_in: torch.Tensor, weight: torch.Tensor) -> bool:
    K, H = weight.shape
    return bool(g_in.shape[0] > 0 and weight.stride(1) == 1 and _lib.lib().tzr_linear_bwd_relu_supported(K, H))


def linear_bwd_relu(g_in: torch.Tensor, weight: torch.Tensor, y: torch.Tensor):
    """((g_in @ weight) * (y > 0), its column sums) in one launch (tzr_linear_bwd_relu): the input gradient of the Linear layer
    with `weight` [K, H] chained with the ReLU mask and bias gradient of the layer that produced y [N, H]."""
    N, K = g_in.shape
    H = weight.shape[1]
    gi, ys = _rows16(g_in), _rows16(y)
    g = torch.empty(N, H, dtype=torch.float32, device=y.device)
    col = torch.empty(H, dtype=torch.float32, device=y.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_linear_bwd_relu_workspace(N, H), y.device)
    _lib.check(L.tzr_linear_bwd_relu(_lib.ptr(gi), gi.stride(0), _lib.ptr(weight), weight.stride(0), _lib.ptr(ys), ys.stride(0), N, K, H,
                                     _lib.ptr(g), g.stride(0), _lib.ptr(col), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(y.device)),
               "tzr_linear_bwd_relu")
    return g, col


# ---- Linear layers over a tall input with the weight resident in registers (csrc/gemm_rows.hip) --------------------------
OWN_ROWS_GEMM = os.environ.get("TZR_OWN_ROWS_GEMM", "1") != "0"  # False: the GEMM library (A/B switch of the bench)
ROWS_GEMM_MIN_ROWS = 4096  # below this many rows a launch of the library's small-tile kernels is as good


def _al16(t: torch.Tensor) -> bool:
    return t.data_ptr() % 16 == 0


def linear_rows_supported(x: torch.Tensor, K: int, H: int) -> bool:
    """x [N, >= K] fp32 with rows 16-byte aligned, (K, H) one of the shapes `tzr_linear_rows` is built for"""
    return bool(OWN_ROWS_GEMM and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] > 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0
                and _al16(x) and _lib.lib().tzr_linear_rows_supported(K, H))


def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False, out_major: bool = True,
                rowvec: Optional[torch.Tensor] = None, row_index: Optional[torch.Tensor] = None, K: Optional[int] = None) -> torch.Tensor:
    """act(x[:, :K] @ W + bias + rowvec[row_index]) in one launch (tzr_linear_rows).  out_major: `weight` is nn.Linear's [H, K]
    (the layer's forward); else `weight` is [K, H] (an input gradient g @ weight).  `K` < x.shape[1] reads the leading columns of
    wider rows."""
    N = x.shape[0]
    if out_major:
        H, Kw = weight.shape
    else:
        Kw, H = weight.shape
    K = Kw if K is None else K
    assert K == Kw and weight.stride(1) == 1
    out = torch.empty(N, H, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tzr_linear_rows(_lib.ptr(x), x.stride(0), _lib.ptr(weight), weight.stride(0), 1 if out_major else 0,
                                          _lib.ptr(bias) if bias is not None else None, _lib.ptr(rowvec) if rowvec is not None else None,
                                          rowvec.stride(0) if rowvec is not None else 0, _lib.ptr(row_index) if row_index is not None else None,
                                          1 if relu else 0, N, K, H, _lib.ptr(out), out.stride(0), _lib.stream_ptr(x.device)), "tzr_linear_rows")
    return out


def linear_rows_wgrad_supported(g: torch.Tensor, x: torch.Tensor, K: Optional[int] = None) -> bool:
    K = x.shape[1] if K is None else K
    ok = lambda t: t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and _al16(t)
    return bool(OWN_ROWS_GEMM and g.shape[0] > 0 and ok(g) and ok(x) and _lib.lib().tzr_linear_rows_wgrad_supported(g.shape[1], K))


def linear_rows_wgrad(g: torch.Tensor, x: torch.Tensor, K: Optional[int] = None) -> torch.Tensor:
    """g^T x[:, :K] ([H, N] x [N, K]): the weight gradient of a Linear layer over a tall input (tzr_linear_rows_wgrad: partial sums
    per workgroup, added in workgroup order -- deterministic)"""
    N, H = g.shape
    K = x.shape[1] if K is None else K
    L = _lib.lib()
    dw = torch.empty(H, K, dtype=torch.float32, device=g.device)
    ws = _lib.workspace(L.tzr_linear_rows_wgrad_workspace(N, H, K), g.device)
    _lib.check(L.tzr_linear_rows_wgrad(_lib.ptr(g), g.stride(0), _lib.ptr(x), x.stride(0), N, H, K, _lib.ptr(dw), dw.stride(0), 0, _lib.ptr(ws),
                                       ws.numel(), _lib.stream_ptr(g.device)), "tzr_linear_rows_wgrad")
    return dw


# ---- small layer stacks as whole-stack kernels (csrc/mlp_ops.hip) -------------------------------------------------
MLP2_MAX = (32, 64, 32)  # input, hidden, output widths tzr_mlp2_* take
TAIL_MAX = (64, 32)      # input, hidden widths tzr_mlp_tail takes


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float()


def weight_grad(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g^T x ([H, B] x [B, K]) -- the weight gradient of a Linear layer.  One output tile of 64 x 783 and a 65536-long
    reduction is the worst shape for a GEMM library: as 16 batched products over slices of the batch plus a 16-way sum
    (fixed order: deterministic) it takes 77 instead of 120 us at B = 65536 (scripts/gemm_variants.py, profiles/r03z)."""
    B = g.shape[0]
    S = 16
    if B >= 16384 and g.is_contiguous() and x.is_contiguous():
        Bm = B // S * S  # (a row count that is no multiple of 16 -- the jagged positions of a sequence batch: the rest as one more product)
        out = torch.bmm(g[:Bm].view(S, Bm // S, g.shape[1]).transpose(1, 2), x[:Bm].view(S, Bm // S, x.shape[1])).sum(0)
        return out if Bm == B else out + g[Bm:].t() @ x[Bm:]
    return g.t() @ x


# (a process-wide counter, not a thread-local: autograd runs the backward of device tensors on its own engine thread,
# which would never see the caller's thread-local -- the first version of this switch did nothing, profiles/r04m)
_ROOT_LOSS = {"on": 0}


@contextlib.contextmanager
def root_loss():
    """`with root_loss(): loss.backward()` (or `autograd.grad(loss, ...)`): the caller differentiates the fused loss
    ITSELF, so the gradient arriving at `_TopLossFn` / `_InteractionTopLossFn` is autograd's 1.0 and the multi-tensor
    launch that scales the six parameter gradients by it (9 us per step) is skipped.  Not for a scaled loss (GradScaler,
    gradient accumulation's 1/steps, a weighted sum of several losses): there the scale is real."""
    _ROOT_LOSS["on"] += 1
    try:
        yield
    finally:
        _ROOT_LOSS["on"] -= 1


_ONES: dict = {}


def unit_gradient(loss: torch.Tensor) -> torch.Tensor:
    """The 1.0 that autograd would create for the root of a backward pass (`ones_like`: a fill launch per step), made once
    per device and dtype: `loss.backward(gradient=unit_gradient(loss))`.  Whoever captures a backward pass into a hipGraph calls
    this once BEFORE the capture (the pipelines do, in their constructors): made inside a capture, the tensor would hold nothing
    until that graph's first replay."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    t = _ONES.get(key)
    if t is None:
        t = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
        _ONES[key] = t
    return t


def _loss_is_root(gl: Optional[torch.Tensor] = None) -> bool:
    """Inside `root_loss()` AND the gradient that arrived IS the cached unit gradient (same storage): a model that scales
    the fused loss before summing (task / sample weights, a GradScaler, gradient accumulation's 1/steps) hands in another
    tensor and takes the scaled path even when its caller announced a root loss."""
    if _ROOT_LOSS["on"] <= 0:
        return False
    if gl is None:
        return True
    one = _ONES.get((gl.device, gl.dtype, tuple(gl.shape)))
    return one is not None and gl.data_ptr() == one.data_ptr()


# ---- gradients left as partial sums for the optimizer's own launch (tzr_dense_adam_fused) ------------------------------------
# With FusedDenseAdam(fuse_finish=True) as the dense optimizer the bottom MLP's backward and the first top-MLP layer's weight
# gradient do not run their finishing launches: the tensors they return to autograd are UNWRITTEN, and what they stand for is
# noted here under the tensor's address; `FusedDenseAdam.step` adds the partial sums up on its way to the parameter update
# (same order of additions: bit-identical) and anything left over is written out by `materialize_pending`.  Who else reads such a
# gradient before the optimizer has run -- clipping, a collective, gradient accumulation -- calls `materialize_pending()` first;
# the flag is the caller's statement that nothing does.
FUSE_FINISH = False
# data_ptr of the returned gradient tensor -> (kind, keep-alive tensors, source fields, generation).  (NO reference to the tensor
# itself: autograd hands a gradient over to `.grad` without a copy only when nobody else holds it -- a copy would be a copy of
# unwritten memory under another address.  Claimed by address: `FusedDenseAdam.step` / `materialize_pending` look their tensors up.)
_PENDING: dict = {}
# id of a parameter whose gradient was left as partial sums -> the backward pass that did it (autograd's graph-task id; cleared by
# the optimizer's step / zero_grad)
_DEFERRED: dict = {}


def _backward_pass_id() -> int:
    f = getattr(torch._C, "_current_graph_task_id", None)  # (guarded: a private hook; without it every pass looks like the same one)
    return int(f()) if f is not None else -1


def _owner_id(params) -> int:
    """id of the fusing optimizer that steps `params` (what _defer_finish just checked): a pending entry remembers whose it is"""
    ent = _FUSED_OWNER.get(id(params[0])) if params else None
    o = ent[1]() if ent is not None else None
    return id(o) if o is not None else 0


def _defer_finish(dev: torch.device, params=()) -> bool:
    """leave this backward's parameter gradients as partial sums?  Only when the flag is up AND none of the parameters holds a
    gradient already: autograd would ADD the new gradient to the old one -- an addition that reads the unwritten tensor."""
    if not FUSE_FINISH or dev.type == "meta":
        return False
    # (the flag is process-wide; the promise is an optimizer's: only for parameters a LIVE FusedDenseAdam(fuse_finish=True) steps)
    for p in params:
        ent = _FUSED_OWNER.get(id(p))
        if ent is None or ent[0]() is not p or ent[1]() is None:
            return False
    held = [p.grad for p in params if getattr(p, "grad", None) is not None]
    if held:
        materialize_pending(held)  # (an earlier backward's gradient, still partial sums: written out before autograd adds to it)
        return False
    task = _backward_pass_id()
    if any(_DEFERRED.get(id(p), None) == task for p in params):  # (an entry of ANOTHER pass: its gradient was dropped unstepped -- overwritten below)
        # a parameter used twice in one backward pass (a shared layer): autograd will ADD the two gradients, and the first one --
        # left as partial sums -- may not have reached `.grad` yet: nothing here can write it out in time.  Never silently.
        raise RuntimeError("FusedDenseAdam(fuse_finish=True): a parameter takes part in the model twice (a shared layer): its two "
                           "gradients would be added before the first is written; construct the optimizer with fuse_finish=False")
    stale = {id(p) for p in params if id(p) in _DEFERRED}
    if stale:  # their earlier pass was abandoned (gradients dropped by hand): what it left behind goes with it
        for ptr in [q for q, e in _PENDING.items() if len(e) > 5 and stale.intersection(e[5])]:
            del _PENDING[ptr]
    _DEFERRED.update((id(p), task) for p in params)
    return True


def _adam_tables(rows):
    """rows: (param | None, grad, exp_avg | None, exp_avg_sq | None, state | None) -> the C arrays of tzr_dense_adam_fused"""
    n = len(rows)
    tab = (_lib.TzrAdamTensor * n)()
    src = (_lib.TzrAdamSource * n)()
    wg = None
    keep = []
    for i, row in enumerate(rows):
        p, gr, m, v, st = row[:5]
        key = row[5] if len(row) > 5 else gr  # the tensor autograd returned (what _PENDING knows); `gr` = where the gradient goes / lies
        tab[i].param = _lib.ptr(p) if p is not None else 0
        tab[i].grad = _lib.ptr(gr)
        tab[i].exp_avg = _lib.ptr(m) if m is not None else 0
        tab[i].exp_avg_sq = _lib.ptr(v) if v is not None else 0
        tab[i].state = _lib.ptr(st) if st is not None else 0
        tab[i].numel = gr.numel()
        pend = _PENDING.pop(key.data_ptr(), None)
        if pend is None:
            src[i].kind = _lib.ADAM_SRC_TENSOR
            if key is not gr:
                src[i].parts = _lib.ptr(key)  # a finished tensor elsewhere: copied
                keep.append((key,))
        elif pend[0] == "rows":
            _, alive, (G, P, col) = pend[:3]
            if col + gr.numel() > P:  # an entry left behind by a tensor that is gone, its address reused: this gradient is a finished tensor
                src[i].kind = _lib.ADAM_SRC_TENSOR
                if key is not gr:
                    src[i].parts = _lib.ptr(key)
                    keep.append((key,))
                continue
            src[i].kind, src[i].G, src[i].P, src[i].col, src[i].parts = _lib.ADAM_SRC_ROWS, G, P, col, _lib.ptr(alive[0])
            keep.append(alive)
        else:
            _, alive, blob = pend[:3]
            if wg is not None:  # (one slice set per launch: the first stays, this one is written out on its own below)
                _PENDING[key.data_ptr()] = pend
                src[i].kind = -1
                continue
            src[i].kind, wg = _lib.ADAM_SRC_WGRAD, blob
            keep.append(alive)
    return tab, src, wg, keep


_OPTIMIZERS: "weakref.WeakSet" = weakref.WeakSet()  # live FusedDenseAdam objects (materialize_pending's default reach)
_FUSED_OWNER: dict = {}  # id(parameter) -> (weakref of the parameter, weakref of the fuse_finish optimizer that steps it)  (by id: tensors compare elementwise)
_GENERATION = [0]  # optimizer steps seen: an entry nobody has claimed two steps later belongs to a tensor that is gone


def materialize_pending(tensors: Optional[Iterable[torch.Tensor]] = None) -> None:
    """Write the gradients among `tensors` (default: the `.grad` of every parameter of every live FusedDenseAdam) that are
    still sets of partial sums out as finished tensors (one launch per 32): for whoever needs them before -- or without --
    `FusedDenseAdam.step`."""
    if not _PENDING:
        return
    if tensors is None:
        tensors = [p.grad for o in list(_OPTIMIZERS) for p in o.params if p.grad is not None]
    todo = [t for t in tensors if t is not None and t.data_ptr() in _PENDING]
    while todo:
        rows, rest, has_wg = [], [], False
        for t in todo:
            is_wg = _PENDING[t.data_ptr()][0] == "wgrad"
            if len(rows) == 32 or (is_wg and has_wg):  # (one slice set per launch)
                rest.append(t)
                continue
            has_wg = has_wg or is_wg
            rows.append((None, t, None, None, None))
        tab, src, wg, keep = _adam_tables(rows)
        _lib.check(_lib.lib().tzr_dense_adam_fused(tab, src, len(rows), C.byref(wg) if wg is not None else None, None, 0.0, 0.9, 0.999,
                                                   1e-8, 0.0, _lib.stream_ptr(rows[0][1].device)), "tzr_dense_adam_fused")
        del keep
        todo = rest


PACKED_LAUNCHES = [0]  # pack_gradients calls that launched / of them: with partial sums among the sources (tests)
PACKED_PARTIALS = [0]


def pack_gradients(grads) -> Optional[torch.Tensor]:
    """torch.cat([g.reshape(-1) for g in grads]) in ONE launch that takes each gradient as it lies -- a finished tensor (copied) or
    the partial sums a backward left for the optimizer (added up on the way, same order as their finishing launch: bit-identical)
    -- the flat buffer of the sharded step's dense all-reduce.  None: not a case for it (the caller concatenates)."""
    grads = list(grads)
    if not grads or len(grads) > 32 or any(g.dtype != torch.float32 or not g.is_contiguous() for g in grads):
        return None
    if sum(1 for g in grads if _PENDING.get(g.data_ptr(), ("",))[0] == "wgrad") > 1:
        return None
    total = sum(g.numel() for g in grads)
    flat = torch.empty(total, dtype=torch.float32, device=grads[0].device)
    rows, o = [], 0
    for g in grads:
        rows.append((None, flat[o:o + g.numel()], None, None, None, g))
        o += g.numel()
    tab, src, wg, keep = _adam_tables(rows)
    if any(e[3] == _GENERATION[0] for e in _PENDING.values()):
        # a backward of THIS step left partial sums that none of `grads` claims: autograd handed its tensor on as a copy (another
        # address) -- the copy is unwritten memory.  Never silently.
        raise RuntimeError("pack_gradients: a gradient left as partial sums (FusedDenseAdam(fuse_finish=True)) is not among the tensors "
                           "to pack -- it reached the caller as a copy; construct the optimizer with fuse_finish=False for this model")
    _lib.check(_lib.lib().tzr_dense_adam_fused(tab, src, len(rows), C.byref(wg) if wg is not None else None, None, 0.0, 0.9, 0.999, 1e-8,
                                               0.0, _lib.stream_ptr(flat.device)), "tzr_dense_adam_fused")
    del keep
    PACKED_LAUNCHES[0] += 1
    PACKED_PARTIALS[0] += int(any(src[i].kind != _lib.ADAM_SRC_TENSOR for i in range(len(rows))))
    return flat


class _Mlp2Fn(torch.autograd.Function):
    """relu(relu(x Wa^T + ba) Wb^T + bb): forward in one launch, the four parameter gradients in one launch + a finish
    (tzr_mlp2_fwd / tzr_mlp2_bwd).  `x` is data (no input gradient): the bottom MLP of DLRM on the dense features."""

    @staticmethod
    def forward(ctx, x, Wa, ba, Wb, bb):
        B, K0 = x.shape
        H1, H2 = Wa.shape[0], Wb.shape[0]
        xs = x if (x.dtype == torch.float32 and x.stride(1) == 1) else x.contiguous().float()
        ha = torch.empty(B, H1, dtype=torch.float32, device=x.device)
        hb = torch.empty(B, H2, dtype=torch.float32, device=x.device)
        Wa_, Wb_, ba_, bb_ = _f32c(Wa), _f32c(Wb), _f32c(ba), _f32c(bb)
        _lib.check(_lib.lib().tzr_mlp2_fwd(_lib.ptr(xs), xs.stride(0), B, K0, _lib.ptr(Wa_), _lib.ptr(ba_), H1, _lib.ptr(Wb_),
                                           _lib.ptr(bb_), H2, _lib.ptr(ha), ha.stride(0), _lib.ptr(hb), hb.stride(0),
                                           _lib.stream_ptr(x.device)), "tzr_mlp2_fwd")
        ctx.save_for_backward(xs, ha, hb, Wb_)
        ctx.dims = (K0, H1, H2)
        ctx.param_refs = (Wa, ba, Wb, bb)  # (backward looks at their `.grad`: see FUSE_FINISH)
        return hb

    @staticmethod
    def backward(ctx, dhb):
        xs, ha, hb, Wb_ = ctx.saved_tensors
        K0, H1, H2 = ctx.dims
        B = xs.shape[0]
        g = _rows16(dhb) if dhb.dtype == torch.float32 else dhb.float().contiguous()
        dWa = torch.empty(H1, K0, dtype=torch.float32, device=xs.device)
        dba = torch.empty(H1, dtype=torch.float32, device=xs.device)
        dWb = torch.empty(H2, H1, dtype=torch.float32, device=xs.device)
        dbb = torch.empty(H2, dtype=torch.float32, device=xs.device)
        L = _lib.lib()
        ws = _lib.workspace(L.tzr_mlp_workspace(), xs.device)
        if _defer_finish(xs.device, ctx.param_refs):
            # no finish launch: the four tensors stay unwritten, the optimizer's launch adds the partial rows up (row layout:
            # [dWb | dbb | dWa | dba], include/tzrec_hip.h)
            G, P = C.c_int(0), C.c_int(0)
            _lib.check(L.tzr_mlp2_bwd_parts(_lib.ptr(g), g.stride(0), _lib.ptr(hb), hb.stride(0), _lib.ptr(ha), ha.stride(0), _lib.ptr(xs),
                                            xs.stride(0), B, K0, H1, H2, _lib.ptr(Wb_), _lib.ptr(ws), ws.numel(), C.byref(G), C.byref(P),
                                            _lib.stream_ptr(xs.device)), "tzr_mlp2_bwd_parts")
            col = 0
            for t in (dWb, dbb, dWa, dba):
                _PENDING[t.data_ptr()] = ("rows", (ws,), (G.value, P.value, col), _GENERATION[0], _owner_id(ctx.param_refs), tuple(id(q) for q in ctx.param_refs))
                col += t.numel()
            return None, dWa, dba, dWb, dbb
        _lib.check(L.tzr_mlp2_bwd(_lib.ptr(g), g.stride(0), _lib.ptr(hb), hb.stride(0), _lib.ptr(ha), ha.stride(0), _lib.ptr(xs),
                                  xs.stride(0), B, K0, H1, H2, _lib.ptr(Wb_), _lib.ptr(dWa), _lib.ptr(dba), _lib.ptr(dWb),
                                  _lib.ptr(dbb), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(xs.device)), "tzr_mlp2_bwd")
        return None, dWa, dba, dWb, dbb


def mlp2(x, Wa, ba, Wb, bb):
    """Two-layer ReLU MLP through the fused kernels; the caller checks `mlp2_fits` first."""
    return _Mlp2Fn.apply(x, Wa, ba, Wb, bb)


def mlp2_fits(x: torch.Tensor, linears) -> bool:
    if len(linears) != 2 or x.dim() != 2 or x.requires_grad or x.shape[0] == 0:
        return False
    a, b = linears
    if a.bias is None or b.bias is None or a.weight.dtype != torch.float32:
        return False
    return a.in_features <= MLP2_MAX[0] and a.out_features <= MLP2_MAX[1] and b.out_features <= MLP2_MAX[2]


def _mlp_tail(y1, labels, W2, b2, w3, b3):
    """tzr_mlp_tail on y1 = relu(first layer): -> (logits, g1, dW2, db2, dw3, scal = {d loss / d b3, loss}, db1)."""
    B, H1 = y1.shape
    H2 = W2.shape[0]
    dev = y1.device
    y = labels.contiguous()
    if y.dtype not in (torch.float32, torch.int32, torch.int64):
        y = y.float()
    logits = torch.empty(B, dtype=torch.float32, device=dev)
    g1 = torch.empty(B, H1, dtype=torch.float32, device=dev)
    dW2 = torch.empty(H2, H1, dtype=torch.float32, device=dev)
    db2 = torch.empty(H2, dtype=torch.float32, device=dev)
    dw3 = torch.empty(1, H2, dtype=torch.float32, device=dev)
    scal = torch.empty(2, dtype=torch.float32, device=dev)
    db1 = torch.empty(H1, dtype=torch.float32, device=dev)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_mlp_workspace(), dev)
    W2_, b2_, w3_, b3_ = _f32c(W2), _f32c(b2), _f32c(w3), _f32c(b3)
    _lib.check(L.tzr_mlp_tail(_lib.ptr(y1), y1.stride(0), _lib.ptr(y), y.element_size(), 1 if y.is_floating_point() else 0,
                              B, H1, _lib.ptr(W2_), _lib.ptr(b2_), H2, _lib.ptr(w3_), _lib.ptr(b3_), _lib.ptr(logits),
                              _lib.ptr(g1), g1.stride(0), _lib.ptr(dW2), _lib.ptr(db2), _lib.ptr(dw3), _lib.ptr(scal),
                              _lib.ptr(db1), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "tzr_mlp_tail")
    return logits, g1, dW2, db2, dw3, scal, db1


class _TopLossFn(torch.autograd.Function):
    """The top MLP of a ranking model from its wide input to the loss, forward AND backward of everything behind the
    first GEMM in one launch: y1 = relu(z W1^T + b1) stays one hipBLASLt call (ReLU in its epilogue); tzr_mlp_tail then
    computes y2, the logit, mean BCE-with-logits and -- the loss being the end of the graph -- the gradients of W2, b2,
    w3, b3, b1 and g1 = d(loss)/d(z W1^T + b1) right away.  backward() only runs the two products of the first layer
    (dz = g1 W1, dW1 = g1^T z), scaled by the incoming gradient of the loss.  Returns (loss, logits); `logits` is for
    predictions / metrics and carries no gradient."""

    @staticmethod
    def forward(ctx, z, W1, b1, W2, b2, w3, b3, labels):
        y1 = torch._addmm_activation(b1, z, W1.t(), use_gelu=False)
        logits, g1, dW2, db2, dw3, scal, db1 = _mlp_tail(y1, labels, W2, b2, w3, b3)
        ctx.save_for_backward(z, W1, g1, dW2, db2, dw3, scal, db1)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)  # (no zero tensor for the gradient of `logits`: a [B] fill per step)
        return scal[1], logits

    @staticmethod
    def backward(ctx, gl, _glogits):
        z, W1, g1, dW2, db2, dw3, scal, db1 = ctx.saved_tensors
        # the incoming gradient of the (scalar) loss scales everything; it is folded into the small operands so the
        # [B, *] tensors are touched by the two GEMMs only
        if _loss_is_root(gl):  # gl is the unit gradient (root_loss): nothing to scale
            dz = (g1 @ W1) if ctx.needs_input_grad[0] else None
            return (dz, weight_grad(g1, z), db1, dW2, db2, dw3, scal[0:1], None)
        dz = (g1 @ (W1 * gl)) if ctx.needs_input_grad[0] else None
        # ... and one multi-tensor launch scales the six parameter gradients (six separate 5 us multiplies otherwise)
        outs = torch._foreach_mul([weight_grad(g1, z), db1, dW2, db2, dw3, scal[0:1]], gl)
        return (dz, *outs, None)


def top_loss_fits(z: torch.Tensor, linears, out_linear) -> bool:
    if len(linears) != 2 or z.dim() != 2 or not z.is_cuda and _lib.backend() != "emu":
        return False
    a, b = linears
    if a.bias is None or b.bias is None or out_linear.bias is None or out_linear.out_features != 1:
        return False
    return a.out_features <= TAIL_MAX[0] and b.out_features <= TAIL_MAX[1] and a.weight.dtype == torch.float32


def top_loss(z, l1, l2, out_linear, labels):
    """(mean BCE-with-logits loss, logits [B]) of relu(relu(z W1^T + b1) W2^T + b2) w3^T + b3 -- see _TopLossFn."""
    return _TopLossFn.apply(z, l1.weight, l1.bias, l2.weight, l2.bias, out_linear.weight, out_linear.bias, labels)


# The weight gradient of the layer behind the interaction: tzr_dot_interaction_top_wgrad (z rebuilt on the chip, nothing kept
# from the forward) or, when False, the GEMM library over a z the forward writes out for it (`weight_grad`; bench.py
# --gemm-wgrad).  Inside the step the two take the same time to within the run-to-run spread at 8 192 .. 65 536 samples
# (profiles/r04ak: 0.1618 / 0.2291 / 0.3536 / 0.5693 ms against 0.1646 / 0.2265 / 0.3564 / 0.5668); the own kernel keeps no
# [B, P + D n] activation (205 MB at 65 536) and leaves no library GEMM in the step.
OWNED_WGRAD = True


def _owned_wgrad(batch: int, row_floats: int) -> bool:
    # (the kernel's sample offsets are 32-bit: B * stride < 2^32 for the embeddings and for g1; a larger batch keeps z)
    return bool(OWNED_WGRAD) and batch * max(row_floats, 64) < (1 << 32)


class _InteractionTopLossFn(torch.autograd.Function):
    """DLRM from the embeddings to the loss: dot interaction + first top-MLP layer as one kernel per direction
    (tzr_dot_interaction_top_fwd / _bwd, csrc/interaction_top.hip), the rest of the top MLP + loss + their backward as
    tzr_mlp_tail.  Neither the interaction row z [B, P + 16 n] nor its gradient dz = g1 W1 ever exists in HBM: the weight
    gradient g1^T z rebuilds z from the embeddings too (tzr_dot_interaction_top_wgrad).  Same returns as _TopLossFn."""

    @staticmethod
    def forward(ctx, dense, sparse, D, W1, b1, W2, b2, w3, b3, labels):
        B = sparse.shape[0]
        F = sparse.shape[1] // D
        sparse = sparse.contiguous()
        dense = dense.contiguous()
        n = F + 1
        width = n * (n - 1) // 2 + D * n
        H1 = W1.shape[0]
        dev = sparse.device
        y1 = torch.empty(B, H1, dtype=torch.float32, device=dev)
        z = None if _owned_wgrad(B, sparse.shape[1]) else torch.empty(B, width, dtype=torch.float32, device=dev)
        W1_, b1_ = _f32c(W1), _f32c(b1)
        _lib.check(_lib.lib().tzr_dot_interaction_top_fwd(
            _lib.ptr(dense), dense.stride(0), _lib.ptr(sparse), sparse.stride(0), F, D, B, _lib.ptr(W1_), W1_.stride(0),
            _lib.ptr(b1_), H1, 1, _lib.ptr(z), width, _lib.ptr(y1), y1.stride(0), _lib.stream_ptr(dev)),
            "tzr_dot_interaction_top_fwd")
        logits, g1, dW2, db2, dw3, scal, db1 = _mlp_tail(y1, labels, W2, b2, w3, b3)
        ctx.save_for_backward(dense, sparse, W1_, g1, dW2, db2, dw3, scal, db1)
        ctx.z = z
        ctx.cfg = (F, D)
        ctx.w1_ref = W1
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)  # (no zero tensor for the gradient of `logits`: a [B] fill per step)
        return scal[1], logits

    @staticmethod
    def backward(ctx, gl, _glogits):
        dense, sparse, W1, g1, dW2, db2, dw3, scal, db1 = ctx.saved_tensors
        F, D = ctx.cfg
        B = sparse.shape[0]
        root = _loss_is_root(gl)  # gl is the unit gradient: no scale operand for the kernel, no multi-tensor scaling launch
        gl32 = None if root else gl.reshape(1).to(torch.float32)
        gd = gs = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gs = torch.empty_like(sparse)
            gd = torch.empty_like(dense)
            _lib.check(_lib.lib().tzr_dot_interaction_top_bwd(
                _lib.ptr(dense), dense.stride(0), _lib.ptr(sparse), sparse.stride(0), F, D, B, _lib.ptr(g1), g1.stride(0),
                g1.shape[1], _lib.ptr(W1), W1.stride(0), _lib.ptr(gl32), _lib.ptr(gd), gd.stride(0), _lib.ptr(gs),
                gs.stride(0), _lib.stream_ptr(sparse.device)), "tzr_dot_interaction_top_bwd")
        if not ctx.needs_input_grad[3]:
            dW1 = None
        elif ctx.z is None:
            dW1 = interaction_top_wgrad(dense, sparse, D, g1, gl32, defer_for=(ctx.w1_ref,))
        else:
            dW1 = weight_grad(g1, ctx.z) if root else weight_grad(g1, ctx.z) * gl
        if root:
            return (gd, gs, None, dW1, db1, dW2, db2, dw3, scal[0:1], None)
        outs = torch._foreach_mul([db1, dW2, db2, dw3, scal[0:1]], gl)
        return (gd, gs, None, dW1, *outs, None)


def interaction_top_wgrad(dense: torch.Tensor, sparse: torch.Tensor, D: int, g1: torch.Tensor,
                          scale: Optional[torch.Tensor] = None, defer_for=None) -> torch.Tensor:
    """dW1 = scale * g1^T z [H, P + D n] of the Linear behind the dot interaction, z rebuilt from (dense, sparse) on the chip
    (tzr_dot_interaction_top_wgrad, csrc/interaction_wgrad.hip): autograd's weight gradient of the first `final_mlp` layer
    (/root/reference/tzrec/modules/mlp.py:58-83 behind models/dlrm.py:123-135) without the [B, P + D n] rows in HBM."""
    B = sparse.shape[0]
    F = sparse.shape[1] // D
    n = F + 1
    width = n * (n - 1) // 2 + D * n
    H = g1.shape[1]
    if B == 0:
        return torch.zeros(H, width, dtype=torch.float32, device=sparse.device)
    dW = torch.empty(H, width, dtype=torch.float32, device=sparse.device)
    L = _lib.lib()
    ws = _lib.workspace(L.tzr_dot_interaction_top_wgrad_workspace(F, D, 1, H), sparse.device)
    if defer_for is not None and _defer_finish(sparse.device, defer_for):  # no slice reduction launch: the optimizer's launch adds the slices up (see FUSE_FINISH)
        blob = _lib.TzrWgradParts()
        _lib.check(L.tzr_dot_interaction_top_wgrad_parts(
            _lib.ptr(dense), dense.stride(0), _lib.ptr(sparse), sparse.stride(0), F, D, B, _lib.ptr(g1), g1.stride(0), H,
            _lib.ptr(scale), _lib.ptr(ws), ws.numel(), C.byref(blob), _lib.stream_ptr(sparse.device)), "tzr_dot_interaction_top_wgrad_parts")
        _PENDING[dW.data_ptr()] = ("wgrad", (ws, scale), blob, _GENERATION[0], _owner_id(defer_for), tuple(id(q) for q in defer_for))
        return dW
    _lib.check(L.tzr_dot_interaction_top_wgrad(
        _lib.ptr(dense), dense.stride(0), _lib.ptr(sparse), sparse.stride(0), F, D, B, _lib.ptr(g1), g1.stride(0), H,
        _lib.ptr(scale), _lib.ptr(dW), dW.stride(0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(sparse.device)),
        "tzr_dot_interaction_top_wgrad")
    return dW


def interaction_top_fits(dense: torch.Tensor, sparse: torch.Tensor, D: int, first_linear) -> bool:
    """Whether the fused interaction + first-layer kernels take this shape (D = 16, H = 64, n <= 32, fp32)."""
    if dense is None or sparse.dim() != 2 or dense.dim() != 2 or dense.shape[1] != D or sparse.shape[1] % D:
        return False
    if sparse.dtype != torch.float32 or dense.dtype != torch.float32 or first_linear.weight.dtype != torch.float32:
        return False
    F = sparse.shape[1] // D
    n = F + 1
    if first_linear.in_features != n * (n - 1) // 2 + D * n or sparse.shape[0] > (1 << 30):  # (samples per call: 32-bit counters)
        return False
    return bool(_lib.lib().tzr_dot_interaction_top_supported(F, D, 1, first_linear.out_features))


def interaction_first_layer(dense, sparse, D, l1) -> torch.Tensor:
    """relu(l1(dot_interaction(dense, sparse))) without the interaction row ever reaching HBM -- inference only
    (no autograd graph is recorded; training goes through interaction_top_loss)."""
    B = sparse.shape[0]
    F = sparse.shape[1] // D
    sparse = sparse.detach().contiguous()
    dense = dense.detach().contiguous()
    W1, b1 = _f32c(l1.weight.detach()), _f32c(l1.bias.detach())
    y1 = torch.empty(B, W1.shape[0], dtype=torch.float32, device=sparse.device)
    _lib.check(_lib.lib().tzr_dot_interaction_top_fwd(
        _lib.ptr(dense), dense.stride(0), _lib.ptr(sparse), sparse.stride(0), F, D, B, _lib.ptr(W1), W1.stride(0), _lib.ptr(b1),
        W1.shape[0], 1, None, 0, _lib.ptr(y1), y1.stride(0), _lib.stream_ptr(sparse.device)), "tzr_dot_interaction_top_fwd")
    return y1


def interaction_top_loss(dense, sparse, D, l1, l2, out_linear, labels):
    """(mean BCE-with-logits loss, logits [B]) of the DLRM head on (dense-MLP output, pooled sparse block):
    top_loss(dot_interaction(dense, sparse), ...) with the interaction row's gradient kept on chip."""
    return _InteractionTopLossFn.apply(dense, sparse, D, l1.weight, l1.bias, l2.weight, l2.bias, out_linear.weight,
                                       out_linear.bias, labels)


class FusedDenseAdam:
    """torch.optim.Adam (amsgrad off) for the dense parameters, ONE launch per step regardless of the
    number of tensors (tzr_dense_adam_fused: a tensor's step count is moved on by the last of its workgroups).  `param_groups[0]["lr"]` may be changed between steps (it is mirrored into a
    device scalar, so a captured hipGraph sees the new value)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, fuse_finish: bool = False) -> None:
        """`fuse_finish`: the backward passes that feed this optimizer leave their gradients as partial sums and `step` adds
        them up inside its own launch (dense.FUSE_FINISH: three launches of the DLRM step fewer).  The caller's statement that
        nothing reads a dense gradient between `loss.backward()` and `step()` (or calls `dense.materialize_pending()` first) and
        that gradients are not accumulated over several backward passes."""
        if fuse_finish:
            global FUSE_FINISH
            FUSE_FINISH = True
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("no parameters")
        _OPTIMIZERS.add(self)
        if fuse_finish:
            me = weakref.ref(self)
            for k in [k for k, e in _FUSED_OWNER.items() if e[0]() is None or e[1]() is None]:
                del _FUSED_OWNER[k]
            for p in self.params:
                _FUSED_OWNER[id(p)] = (weakref.ref(p), me)
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
                raise ValueError("FusedDenseAdam needs contiguous float32 parameters on one device")
        self.param_groups = [{"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": float(weight_decay),
                              "params": self.params}]
        self.device = dev
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        # per tensor: [0] step count, [1 ..] arrival counters of tzr_dense_adam_fused (TZR_ADAM_FUSED_STATE floats, zero between launches)
        self._state = torch.zeros(len(self.params), 40, dtype=torch.float32, device=dev)
        self._lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=dev)
        self._lr_host = float(lr)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            _DEFERRED.pop(id(p), None)
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
        elif g["lr"] != self._lr_host:  #  a capture would freeze the stale rate into the graph
            raise RuntimeError("learning rate changed since the last sync and a hipGraph capture is open: call "
                               "dense.sync_learning_rates(model, optimizer) before capturing / replaying (INTEGRATION.md)")
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
        b1, b2 = g["betas"]
        for base in range(0, len(rows), 32):
            tab, src, wg, keep = _adam_tables([(p.data, gr, m, v, st) for p, gr, m, v, st in rows[base:base + 32]])
            if any(src[i].kind < 0 for i in range(len(tab))):  # (a second slice set in one launch: written out first)
                materialize_pending([gr for _, gr, _, _, _ in rows[base:base + 32]])
                tab, src, wg, keep = _adam_tables([(p.data, gr, m, v, st) for p, gr, m, v, st in rows[base:base + 32]])
            _lib.check(_lib.lib().tzr_dense_adam_fused(tab, src, len(tab), C.byref(wg) if wg is not None else None, _lib.ptr(self._lr_dev),
                                                       g["lr"], b1, b2, g["eps"], g["weight_decay"], _lib.stream_ptr(self.device)),
                       "tzr_dense_adam_fused")
            del keep
        for p in self.params:
            _DEFERRED.pop(id(p), None)
        if any(len(e) > 4 and e[4] == id(self) and e[3] == _GENERATION[0] for e in _PENDING.values()):
            # a backward left a gradient of one of THIS optimizer's parameters as partial sums and the tensor that reached `.grad` is
            # another one (autograd copied it): its parameter was just stepped with unwritten memory.  Never silently.
            raise RuntimeError("FusedDenseAdam(fuse_finish=True): a gradient left as partial sums did not reach its parameter's .grad "
                               "as the tensor the backward returned (it was copied on the way); construct the optimizer with "
                               "fuse_finish=False for this model")
        if _PENDING:  # (entries of tensors that are gone -- a gradient autograd dropped: unclaimed two steps later)
            _GENERATION[0] += 1
            for ptr in [q for q, e in _PENDING.items() if e[3] < _GENERATION[0] - 2]:
                del _PENDING[ptr]

    def state_dict(self) -> dict:
        return {"state": {i: {"step": self._state[i, 0].clone(), "exp_avg": m, "exp_avg_sq": v}
                          for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))},
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd: dict) -> None:
        self._state[:, 1:].zero_()  # (tzr_dense_adam_fused's arrival counters: zero between launches)
        for i, st in sd["state"].items():
            self.exp_avg[int(i)].copy_(st["exp_avg"])
            self.exp_avg_sq[int(i)].copy_(st["exp_avg_sq"])
            self._state[int(i), 0] = float(st["step"])  # (the bias corrections are computed from it inside the launch)
        for k, v in sd["param_groups"][0].items():
            self.param_groups[0][k] = v


def lr_sync_targets(model, optimizer=None) -> list:
    """Every object with a `sync_lr()` reachable from the model's collections (their fused sparse optimizers) and from the
    dense optimizer behind its wrappers (`_optimizer` chains, `_optims` / `optimizers` lists of combined optimizers) --
    collected ONCE; a train step iterates the list (walking `model.modules()` every step was host time on the
    hot path, and a combined optimizer's sub-optimizers were never reached)."""
    out, seen = [], set()

    def add(o):
        if o is not None and id(o) not in seen and hasattr(o, "sync_lr"):
            seen.add(id(o))
            out.append(o)

    for m in (list(model.modules()) if hasattr(model, "modules") else []):
        try:
            add(getattr(m, "fused_optimizer", None))
        except Exception:
            pass
        for lane in ("local", "replica"):  # the two halves of a sharded collection hold their own
            sub = getattr(m, lane, None)
            if sub is not None:
                add(getattr(sub, "fused_optimizer", None))
    stack, hops = [optimizer], 0
    while stack and hops < 64:
        opt = stack.pop()
        hops += 1
        if opt is None:
            continue
        add(opt)
        stack.append(getattr(opt, "_optimizer", None))
        for name in ("_optims", "optimizers"):
            subs = getattr(opt, name, None)
            if isinstance(subs, (list, tuple)):
                stack.extend(o[1] if isinstance(o, tuple) else o for o in subs)
            elif isinstance(subs, dict):
                stack.extend(subs.values())
    return out


def sync_learning_rates(model, optimizer=None, targets: Optional[list] = None) -> None:
    """Before replaying a hipGraph that contains update kernels: copy every learning rate a scheduler may have changed
    (fused sparse optimizers of the model's collections, the dense optimizer behind its wrappers) into the device
    scalars the captured kernels read.  No-ops when nothing changed; never call it under capture.  `targets`: the cached
    result of `lr_sync_targets` (train steps collect it once)."""
    for o in (targets if targets is not None else lr_sync_targets(model, optimizer)):
        o.sync_lr()
