"""`torch.library` registration of the C-ABI entry points: `torch.ops.tzrec_hip.*`.

SURVEY.md 8(b), last cell: the kernel-level seam of the reference is `torch.ops.fbgemm.*`
(/root/reference/tzrec/optim/optimizer.py:16-20,258 is where tzrec itself touches it), i.e. ops a
tracer can see.  A bare ctypes call is opaque to FX / `torch.export`, which tzrec's export path
traces -- so every hot-path entry point of `include/tzrec_hip.h` is also an op with a fake-tensor
(meta) implementation; the differentiable ones carry their backward op:

    tzrec_hip::dot_interaction_fwd / dot_interaction_bwd      (autograd registered)
    tzrec_hip::fm_fwd / fm_bwd                                (autograd registered)
    tzrec_hip::kjt_permute, tzrec_hip::block_bucketize        (integer index stage)
    tzrec_hip::pooled_fwd                                      (K5 + K8)
    tzrec_hip::pooled_bwd_adagrad / pooled_bwd_rowwise_adagrad (K6 + K7, mutate weights and state)

The real implementations call the same library as the modules do (`_lib`): there is one backend.
The pooled ops take the descriptor arrays the modules upload (`EmbeddingBagCollection._meta`) as
uint8 tensors plus the table tensors themselves, so the data dependence on the weights is visible
to a tracer (the kernels reach them through the addresses in the descriptors).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib

_NS = "tzrec_hip"


def _width(n_sparse: int, has_dense: bool, dim: int, cat_dense: bool, cat_sparse: bool) -> int:
    n = n_sparse + (1 if has_dense else 0)
    return n * (n - 1) // 2 + (dim if (cat_dense and has_dense) else 0) + (n_sparse * dim if cat_sparse else 0)


# ---- dot interaction -------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::dot_interaction_fwd", mutates_args=())
def dot_interaction_fwd(dense: Optional[Tensor], sparse: Tensor, dim: int, cat_dense: bool, cat_sparse: bool) -> Tensor:
    B, F = sparse.shape[0], sparse.shape[1] // dim
    sparse = sparse.contiguous()
    dense = dense.contiguous() if dense is not None else None
    out = torch.empty(B, _width(F, dense is not None, dim, cat_dense, cat_sparse), dtype=torch.float32, device=sparse.device)
    rc = _lib.lib().tzr_dot_interaction_fwd(
        _lib.ptr(dense), dense.stride(0) if dense is not None else 0, _lib.ptr(sparse), sparse.stride(0), F, dim, B,
        _lib.ptr(out), out.stride(0), int(cat_dense), int(cat_sparse), _lib.stream_ptr(sparse.device))
    _lib.check(rc, "tzr_dot_interaction_fwd")
    return out


@dot_interaction_fwd.register_fake
def _(dense, sparse, dim, cat_dense, cat_sparse):
    return sparse.new_empty(sparse.shape[0], _width(sparse.shape[1] // dim, dense is not None, dim, cat_dense, cat_sparse),
                            dtype=torch.float32)


@torch.library.custom_op(f"{_NS}::dot_interaction_bwd", mutates_args=())
def dot_interaction_bwd(dense: Optional[Tensor], sparse: Tensor, gout: Tensor, dim: int, cat_dense: bool,
                        cat_sparse: bool) -> Tuple[Tensor, Tensor]:
    """(grad dense -- empty [0] when there is no dense row --, grad sparse)"""
    B, F = sparse.shape[0], sparse.shape[1] // dim
    sparse, gout = sparse.contiguous(), gout.contiguous()
    dense = dense.contiguous() if dense is not None else None
    gs = torch.empty_like(sparse)
    gd = torch.empty_like(dense) if dense is not None else sparse.new_empty(0)
    rc = _lib.lib().tzr_dot_interaction_bwd(
        _lib.ptr(dense), dense.stride(0) if dense is not None else 0, _lib.ptr(sparse), sparse.stride(0), F, dim, B,
        _lib.ptr(gout), gout.stride(0), int(cat_dense), int(cat_sparse), _lib.ptr(gd) if dense is not None else None,
        gd.stride(0) if dense is not None else 0, _lib.ptr(gs), gs.stride(0), _lib.stream_ptr(sparse.device))
    _lib.check(rc, "tzr_dot_interaction_bwd")
    return gd, gs


@dot_interaction_bwd.register_fake
def _(dense, sparse, gout, dim, cat_dense, cat_sparse):
    return (torch.empty_like(dense) if dense is not None else sparse.new_empty(0)), torch.empty_like(sparse)


def _di_setup(ctx, inputs, output):
    dense, sparse, dim, cat_dense, cat_sparse = inputs
    ctx.save_for_backward(dense, sparse)
    ctx.cfg = (dim, cat_dense, cat_sparse)


def _di_backward(ctx, gout):
    dense, sparse = ctx.saved_tensors
    dim, cat_dense, cat_sparse = ctx.cfg
    gd, gs = torch.ops.tzrec_hip.dot_interaction_bwd(dense, sparse, gout, dim, cat_dense, cat_sparse)
    return (gd if dense is not None else None), gs, None, None, None


dot_interaction_fwd.register_autograd(_di_backward, setup_context=_di_setup)


# ---- factorization machine -------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::fm_fwd", mutates_args=())
def fm_fwd(x: Tensor) -> Tensor:
    B, F, D = x.shape
    x = x.contiguous()
    out = torch.empty(B, D, dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().tzr_fm_fwd(_lib.ptr(x), x.stride(0), F, D, B, _lib.ptr(out), out.stride(0), _lib.stream_ptr(x.device)),
               "tzr_fm_fwd")
    return out


@fm_fwd.register_fake
def _(x):
    return x.new_empty(x.shape[0], x.shape[2], dtype=torch.float32)


@torch.library.custom_op(f"{_NS}::fm_bwd", mutates_args=())
def fm_bwd(x: Tensor, gout: Tensor) -> Tensor:
    B, F, D = x.shape
    x, gout = x.contiguous(), gout.contiguous()
    gx = torch.empty_like(x)
    _lib.check(_lib.lib().tzr_fm_bwd(_lib.ptr(x), x.stride(0), F, D, B, _lib.ptr(gout), gout.stride(0), _lib.ptr(gx), gx.stride(0),
                                     _lib.stream_ptr(x.device)), "tzr_fm_bwd")
    return gx


@fm_bwd.register_fake
def _(x, gout):
    return torch.empty_like(x)


fm_fwd.register_autograd(lambda ctx, g: torch.ops.tzrec_hip.fm_bwd(ctx.saved_tensors[0], g),
                         setup_context=lambda ctx, inputs, output: ctx.save_for_backward(inputs[0]))


# ---- index stage -----------------------------------------------------------------------------------
@torch.library.custom_op(f"{_NS}::kjt_permute", mutates_args=())
def kjt_permute(permute: Tensor, lengths: Tensor, offsets: Tensor, values: Tensor, weights: Optional[Tensor], n_in_keys: int,
                stride: int, n_out_max: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """K1 (fbgemm permute_2D_sparse_data): (out lengths, out offsets, out values[n_out_max], out weights).
    Output key t takes input key permute[t]; the permuted total is out_offsets[-1] <= n_out_max."""
    dev, T, B = values.device, permute.numel(), stride
    L = _lib.lib()
    out_len = torch.empty(T * B, dtype=lengths.dtype, device=dev)
    out_off = torch.empty(T * B + 1, dtype=torch.int64, device=dev)
    out_val = torch.empty(n_out_max, dtype=torch.int64, device=dev)
    out_w = torch.empty(n_out_max if weights is not None else 0, dtype=torch.float32, device=dev)
    ws = _lib.workspace(L.tzr_kjt_permute_workspace(T, B), dev)
    rc = L.tzr_kjt_permute(_lib.ptr(permute), T, n_in_keys, B, _lib.ptr(lengths), lengths.element_size(), _lib.ptr(offsets),
                           _lib.ptr(values), _lib.ptr(weights), _lib.ptr(out_len), _lib.ptr(out_off), _lib.ptr(out_val),
                           _lib.ptr(out_w) if weights is not None else None, n_out_max, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "tzr_kjt_permute")
    return out_len, out_off, out_val, out_w


@kjt_permute.register_fake
def _(permute, lengths, offsets, values, weights, n_in_keys, stride, n_out_max):
    T = permute.numel()
    return (lengths.new_empty(T * stride), offsets.new_empty(T * stride + 1), values.new_empty(n_out_max),
            values.new_empty(n_out_max if weights is not None else 0, dtype=torch.float32))


@torch.library.custom_op(f"{_NS}::block_bucketize", mutates_args=())
def block_bucketize(block_sizes: Tensor, rank_offsets: Optional[Tensor], lengths: Tensor, offsets: Tensor, values: Tensor,
                    weights: Optional[Tensor], stride: int, world: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """K2 (fbgemm block_bucketize_sparse_features, row-wise): (new lengths [W*F*B] rank-major, new offsets,
    new values (local row ids), new weights, unbucketize permute)."""
    dev, F, B, N = values.device, block_sizes.numel(), stride, values.numel()
    L = _lib.lib()
    new_len = torch.empty(world * F * B, dtype=lengths.dtype, device=dev)
    new_off = torch.empty(world * F * B + 1, dtype=torch.int64, device=dev)
    new_val = torch.empty(N, dtype=torch.int64, device=dev)
    new_w = torch.empty(N if weights is not None else 0, dtype=torch.float32, device=dev)
    unb = torch.empty(N, dtype=torch.int64, device=dev)
    ws = _lib.workspace(L.tzr_block_bucketize_workspace(F, B, world), dev)
    rc = L.tzr_block_bucketize(_lib.ptr(block_sizes), _lib.ptr(rank_offsets), F, B, world, _lib.ptr(offsets), _lib.ptr(values),
                               _lib.ptr(weights), N, _lib.ptr(new_len), new_len.element_size(), _lib.ptr(new_off), _lib.ptr(new_val),
                               _lib.ptr(new_w) if weights is not None else None, _lib.ptr(unb), _lib.ptr(ws), ws.numel(),
                               _lib.stream_ptr(dev))
    _lib.check(rc, "tzr_block_bucketize")
    return new_len, new_off, new_val, new_w, unb


@block_bucketize.register_fake
def _(block_sizes, rank_offsets, lengths, offsets, values, weights, stride, world):
    F, N = block_sizes.numel(), values.numel()
    return (lengths.new_empty(world * F * stride), offsets.new_empty(world * F * stride + 1), values.new_empty(N),
            values.new_empty(N if weights is not None else 0, dtype=torch.float32), values.new_empty(N))


# ---- pooled lookup + fused backward ----------------------------------------------------------------
def _dsts(tensors: List[Tensor]):
    arr = (_lib.TzrDst * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i].ptr, arr[i].stride = _lib.ptr(t), t.stride(0)
    return arr


@torch.library.custom_op(f"{_NS}::pooled_fwd", mutates_args=())
def pooled_fwd(tables: List[Tensor], d_tables: Tensor, d_feats: Tensor, d_slots: Tensor, values: Tensor, offsets: Optional[Tensor],
               weights: Optional[Tensor], stride: int, dst_widths: List[int], mixed_dtype: bool) -> List[Tensor]:
    """K5 + K8 (TBE forward + permute_pooled_embs / regroup): one [B, width] tensor per feature group.
    `tables` = the table tensors the descriptors point into (data dependence only); `offsets` None =
    every bag holds exactly one id."""
    dev, B = values.device, stride
    outs = [torch.empty(B, w, dtype=torch.float32, device=dev) for w in dst_widths]
    rc = _lib.lib().tzr_pooled_fwd_ex(_lib.ptr(d_tables), _lib.ptr(d_feats), d_feats.numel() // _lib.FEATURE_DT.itemsize, _lib.ptr(d_slots),
                                      d_slots.numel() // _lib.SLOT_DT.itemsize, _lib.ptr(values), _lib.ptr(offsets), _lib.ptr(weights), B,
                                      _dsts(outs), len(outs), 1 if offsets is None else 0, _lib.FWD_MIXED_DTYPE if mixed_dtype else 0,
                                      _lib.stream_ptr(dev))
    _lib.check(rc, "tzr_pooled_fwd")
    return outs


@pooled_fwd.register_fake
def _(tables, d_tables, d_feats, d_slots, values, offsets, weights, stride, dst_widths, mixed_dtype):
    return [values.new_empty(stride, w, dtype=torch.float32) for w in dst_widths]


def _pooled_bwd(kind: int, tables, states, d_tables, d_feats, values, offsets, weights, grads, lr, stride, n_keys, max_rows, eps):
    dev, L = values.device, _lib.lib()
    B, N = stride, values.numel()
    F, T = d_feats.numel() // _lib.FEATURE_DT.itemsize, d_tables.numel() // _lib.TABLE_DT.itemsize
    uniform = offsets is None
    # capacity of the table-major position space.  Ragged bags: a key read through k tables contributes its ids k times
    # (EmbeddingBagCollection._n_positions knows k from its lookups: N * max k); the op sees only device descriptors, so it
    # takes the bound that holds for every sharing pattern, N * F -- F times the module's workspace when no key is shared
    # (ADVICE round 2 suggested N: too small as soon as two tables read one key).
    NP = F * B if uniform else N * F
    max_dim = max(int(t.shape[1]) for t in tables)
    ws = _lib.workspace(L.tzr_pooled_bwd_workspace(N, NP, F, T, B, max_dim), dev)
    rc = L.tzr_pooled_bwd_plan(_lib.ptr(d_tables), T, _lib.ptr(d_feats), F, n_keys, max_rows, max_dim, _lib.ptr(values), _lib.ptr(offsets),
                               N, NP, B, 1 if uniform else 0, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "tzr_pooled_bwd_plan")
    opt = _lib.TzrSparseOptim()
    opt.kind, opt.weight_decay_mode, opt.d_lr, opt.eps = kind, _lib.WD_NONE, _lib.ptr(lr), eps
    gl = [g.contiguous() for g in grads]
    rc = L.tzr_pooled_bwd_apply(_lib.ptr(d_tables), _lib.ptr(d_feats), F, T, max_dim, _lib.ptr(offsets), _lib.ptr(weights), N, NP, B,
                                1 if uniform else 0, 0, _dsts(gl), len(gl), opt, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "tzr_pooled_bwd_apply")


@torch.library.custom_op(f"{_NS}::pooled_bwd_adagrad", mutates_args=("tables", "states"))
def pooled_bwd_adagrad(tables: List[Tensor], states: List[Tensor], d_tables: Tensor, d_feats: Tensor, values: Tensor,
                       offsets: Optional[Tensor], weights: Optional[Tensor], grads: List[Tensor], lr: Tensor, stride: int, n_keys: int,
                       max_rows: int, eps: float) -> None:
    """K6 + K7 with elementwise Adagrad fused (fbgemm split_embedding_backward_codegen_adagrad_*_exact):
    updates `tables` and `states` in place; `grads` mirror pooled_fwd's outputs; `lr` is a device scalar."""
    _pooled_bwd(_lib.OPT_ADAGRAD, tables, states, d_tables, d_feats, values, offsets, weights, grads, lr, stride, n_keys, max_rows, eps)


@torch.library.custom_op(f"{_NS}::pooled_bwd_rowwise_adagrad", mutates_args=("tables", "states"))
def pooled_bwd_rowwise_adagrad(tables: List[Tensor], states: List[Tensor], d_tables: Tensor, d_feats: Tensor, values: Tensor,
                               offsets: Optional[Tensor], weights: Optional[Tensor], grads: List[Tensor], lr: Tensor, stride: int,
                               n_keys: int, max_rows: int, eps: float) -> None:
    """K6 + K7 with row-wise Adagrad fused (one state scalar per row)."""
    _pooled_bwd(_lib.OPT_ROWWISE_ADAGRAD, tables, states, d_tables, d_feats, values, offsets, weights, grads, lr, stride, n_keys,
                max_rows, eps)


@pooled_bwd_adagrad.register_fake
def _(tables, states, d_tables, d_feats, values, offsets, weights, grads, lr, stride, n_keys, max_rows, eps):
    return None


@pooled_bwd_rowwise_adagrad.register_fake
def _(tables, states, d_tables, d_feats, values, offsets, weights, grads, lr, stride, n_keys, max_rows, eps):
    return None


OPS = ("dot_interaction_fwd", "dot_interaction_bwd", "fm_fwd", "fm_bwd", "kjt_permute", "block_bucketize", "pooled_fwd",
       "pooled_bwd_adagrad", "pooled_bwd_rowwise_adagrad")
