"""Feature-interaction modules on the gfx950 kernels.

``InteractionArch`` and ``FactorizationMachine`` keep the reference's constructor and ``forward``
(/root/reference/tzrec/modules/interaction.py:57-91, /root/reference/tzrec/modules/fm.py:17-42).
``dot_interaction`` is the fused form DLRM uses here: it reads the dense-MLP output and the pooled
sparse block separately (no ``cat`` to build [B, 27, 16]) and writes
``[interactions | dense | sparse]`` in one pass (/root/reference/tzrec/models/dlrm.py:123-130).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib
from . import ops as _ops  # noqa: F401  (registers torch.ops.tzrec_hip.*)


def _traced(t: torch.Tensor) -> bool:
    """True when `t` is being traced (FX / make_fx / dynamo / torch.export: fake or proxy tensors).  Traced
    programs go through `torch.ops.tzrec_hip.*` so the graph shows the ops; eager calls keep
    the direct autograd.Function path below, which costs no dispatcher round trip per call."""
    return torch.compiler.is_compiling() or type(t) not in (torch.Tensor, nn.Parameter)


class _DotInteractionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense: Optional[torch.Tensor], sparse: torch.Tensor, D: int, cat_dense: bool, cat_sparse: bool):
        B = sparse.shape[0]
        F = sparse.shape[1] // D
        sparse = sparse.contiguous()
        if dense is not None:
            dense = dense.contiguous()
        n = F + (1 if dense is not None else 0)
        width = n * (n - 1) // 2 + (D if (cat_dense and dense is not None) else 0) + (F * D if cat_sparse else 0)
        out = torch.empty(B, width, dtype=torch.float32, device=sparse.device)
        rc = _lib.lib().tzr_dot_interaction_fwd(
            _lib.ptr(dense), dense.stride(0) if dense is not None else 0, _lib.ptr(sparse),
            sparse.stride(0), F, D, B, _lib.ptr(out), out.stride(0), int(cat_dense), int(cat_sparse),
            _lib.stream_ptr(sparse.device),
        )
        _lib.check(rc, "tzr_dot_interaction_fwd")
        ctx.save_for_backward(dense, sparse)
        ctx.cfg = (F, D, cat_dense, cat_sparse)
        return out

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        dense, sparse = ctx.saved_tensors
        F, D, cat_dense, cat_sparse = ctx.cfg
        B = sparse.shape[0]
        gout = gout.contiguous()
        gs = torch.empty_like(sparse)
        gd = torch.empty_like(dense) if dense is not None else None
        rc = _lib.lib().tzr_dot_interaction_bwd(
            _lib.ptr(dense), dense.stride(0) if dense is not None else 0, _lib.ptr(sparse),
            sparse.stride(0), F, D, B, _lib.ptr(gout), gout.stride(0), int(cat_dense),
            int(cat_sparse), _lib.ptr(gd), gd.stride(0) if gd is not None else 0, _lib.ptr(gs),
            gs.stride(0), _lib.stream_ptr(sparse.device),
        )
        _lib.check(rc, "tzr_dot_interaction_bwd")
        return gd, gs, None, None, None


def dot_interaction(
    dense: Optional[torch.Tensor], sparse: torch.Tensor, dim: int, cat_dense: bool = True, cat_sparse: bool = True
) -> torch.Tensor:
    """[B, n(n-1)/2 (+dim) (+F*dim)]: strict-upper-triangle of X X^T for X = [dense; sparse rows]."""
    if _traced(sparse):
        return torch.ops.tzrec_hip.dot_interaction_fwd(dense, sparse, dim, cat_dense, cat_sparse)
    return _DotInteractionFn.apply(dense, sparse, dim, cat_dense, cat_sparse)


class InteractionArch(nn.Module):
    """Feature interaction module (same signature as the reference: ``feature_num``; input
    ``B x N x D``; output ``B x N(N-1)/2``)."""

    def __init__(self, feature_num: int) -> None:
        super().__init__()
        self.feature_num = feature_num

    def output_dim(self) -> int:
        return self.feature_num * (self.feature_num - 1) // 2

    def forward(self, features: torch.Tensor) -> torch.Tensor:
        B, N, D = features.shape
        return dot_interaction(None, features.reshape(B, N * D), D, False, False)


class _FMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor):
        B, F, D = x.shape
        x = x.contiguous()
        out = torch.empty(B, D, dtype=torch.float32, device=x.device)
        rc = _lib.lib().tzr_fm_fwd(_lib.ptr(x), x.stride(0), F, D, B, _lib.ptr(out), out.stride(0), _lib.stream_ptr(x.device))
        _lib.check(rc, "tzr_fm_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        (x,) = ctx.saved_tensors
        B, F, D = x.shape
        gout = gout.contiguous()
        gx = torch.empty_like(x)
        rc = _lib.lib().tzr_fm_bwd(
            _lib.ptr(x), x.stride(0), F, D, B, _lib.ptr(gout), gout.stride(0), _lib.ptr(gx), gx.stride(0), _lib.stream_ptr(x.device)
        )
        _lib.check(rc, "tzr_fm_bwd")
        return gx


class FactorizationMachine(nn.Module):
    """FM second-order term: [B, N, D] -> [B, D] (same signature as the reference)."""

    def forward(self, feature: torch.Tensor) -> torch.Tensor:
        if _traced(feature):
            return torch.ops.tzrec_hip.fm_fwd(feature)
        return _FMFn.apply(feature)
