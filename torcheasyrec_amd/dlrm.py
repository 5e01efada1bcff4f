"""DLRM / DeepFM glue around the gfx950 embedding + interaction kernels.

Mirrors the forward of the reference models:
/root/reference/tzrec/models/dlrm.py:101-135, /root/reference/tzrec/models/deepfm.py:72-108,
/root/reference/tzrec/modules/mlp.py:21-177 (Linear + ReLU per hidden unit),
/root/reference/tzrec/models/rank_model.py:133-262 (logits -> BCEWithLogitsLoss, mean).
The MLPs stay on PyTorch (hipBLASLt); everything sparse goes through libtzrec_hip.so.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from .embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
from .interaction import FactorizationMachine, dot_interaction
from .sparse import KeyedJaggedTensor


import os

_GEMV_OUTPUT = os.environ.get("TZR_OUTPUT_GEMV", "1") == "1"  # A/B switch (name kept from the gemv experiment)
_FUSED_HEAD_BWD = os.environ.get("TZR_FUSED_HEAD_BWD", "1") == "1"  # A/B switch
_FUSED_RELU_BWD = os.environ.get("TZR_MLP_FUSED_RELU_BWD", "1") == "1"  # A/B switch
_FUSED_RELU = os.environ.get("TZR_MLP_FUSED_RELU", "1") == "1"  # A/B switch; measured -23 us per DLRM step
_FUSED_MLP2 = os.environ.get("TZR_FUSED_MLP2", "1") == "1"  # A/B switch: bottom MLP through tzr_mlp2_fwd / bwd
_FUSED_TOP_LOSS = os.environ.get("TZR_FUSED_TOP_LOSS", "1") == "1"  # A/B switch: DLRM.forward_loss through tzr_mlp_tail
# A/B switch: interaction + first top layer as one kernel per direction (csrc/interaction_top.hip); below MIN_B samples the
# persistent kernels' prologue (W1 into registers, 200 KB per workgroup) is not amortised
_FUSED_IA_TOP = os.environ.get("TZR_FUSED_IA_TOP", "1") == "1"
_FUSED_IA_TOP_MIN_B = int(os.environ.get("TZR_FUSED_IA_TOP_MIN_B", "0"))


def _on_emulator() -> bool:
    from . import _lib

    return _lib._lib is not None and _lib.backend() == "emu"



class _LinearReluFn(torch.autograd.Function):
    """relu(x W^T + b) with the ReLU in the GEMM epilogue (torch._addmm_activation has no autograd
    formula in this PyTorch build, so the three backward products are spelled out)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = torch._addmm_activation(bias, x, weight.t(), use_gelu=False)
        ctx.save_for_backward(x, weight, y)
        ctx.bias_ref = bias  # (the Parameter: dense._defer_finish looks at who steps it)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        if _FUSED_RELU_BWD and y.shape[1] % 4 == 0 and y.shape[1] <= 1024 and gy.dtype == torch.float32:
            from .dense import relu_bwd_colsum

            g, gb = relu_bwd_colsum(gy, y, defer_for=(ctx.bias_ref,))  # mask + bias gradient in one pass (tzr_relu_bwd_colsum)
        else:
            g = torch.ops.aten.threshold_backward(gy, y, 0.0)
            gb = g.sum(0)
        gx = g @ weight if ctx.needs_input_grad[0] else None
        from .dense import weight_grad

        return gx, weight_grad(g, x), gb


class _Linear1Fn(torch.autograd.Function):
    """Linear with ONE output unit.  Autograd's weight gradient for nn.Linear is gy^T @ x, a
    [1, B] x [B, in] GEMM: one output tile and a 65536-long reduction -- 68 us on hipBLASLt at
    B = 65536 for 8 MB of input (rocBLAS gemv is worse: ~380 us).  A broadcast multiply and a column
    sum do it in two short launches."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        if _FUSED_HEAD_BWD and x.shape[1] % 4 == 0 and x.shape[1] <= 1024 and x.dtype == torch.float32:
            from .dense import head_bwd

            return head_bwd(gy, x, weight, ctx.needs_input_grad[0])  # one pass over x (tzr_head_bwd)
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        return gx, (x * gy).sum(0, keepdim=True), gy.sum(0)


class _SkinnyLinearFn(torch.autograd.Function):
    """Linear with <= 8 output units (logits, MMoE gates) as two one-pass kernels instead of GEMMs that are one output tile
    wide (tzr_skinny_linear_fwd / _bwd, csrc/dense_ops.hip: 31 us per forward GEMM at [8192, 64] x [64, 1], profiles/r05j)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        from .dense import skinny_linear_fwd

        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.param_refs = (weight, bias) if bias is not None else None  # (dense._defer_finish looks at who steps them)
        return skinny_linear_fwd(x, weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, gy):
        from .dense import skinny_linear_bwd

        x, weight = ctx.saved_tensors
        gx, gw, gb = skinny_linear_bwd(gy.float(), x, weight, ctx.needs_input_grad[0], defer_for=ctx.param_refs)
        return gx, gw, (gb if ctx.has_bias else None)


_SKINNY_LINEAR = os.environ.get("TZR_SKINNY_LINEAR", "1") == "1"  # A/B switch


class OutputLinear(nn.Linear):
    """A Linear layer with a handful of output units -- the logits layer, an MMoE gate -- (same parameters / state_dict keys
    as nn.Linear)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _SKINNY_LINEAR and (x.is_cuda or _on_emulator()) and x.dim() == 2:
            from .dense import skinny_linear_ok

            if skinny_linear_ok(x, self.weight):
                return _SkinnyLinearFn.apply(x, self.weight, self.bias)
        if _GEMV_OUTPUT and self.out_features == 1 and x.is_cuda and x.dim() == 2 and self.bias is not None:
            return _Linear1Fn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _activation(name: Optional[str]) -> Optional[nn.Module]:
    """`activation` of the MLP proto (tzrec/protos/module.proto:9-10; tzrec/modules/activation.py
    create_activation): "nn.ReLU" style names of torch.nn; empty = none."""
    if not name:
        return None
    cls = getattr(nn, name[3:], None) if name.startswith("nn.") else None
    if cls is None:
        raise NotImplementedError(f"MLP activation {name!r}: only torch.nn activations (\"nn.ReLU\", \"nn.GELU\", ...) are built")
    return cls()


class MLP(nn.Module):
    """Stack of Perceptrons (tzrec/modules/mlp.py:21-177): per layer Linear -> [BatchNorm1d | LayerNorm]
    -> activation -> [Dropout]; with `use_bn` the Linear has no bias (mlp.py:60).  The reference
    defaults (bias, ReLU, no norm, no dropout) are the DLRM / DeepFM case and take the fused path."""

    def __init__(self, in_features: int, hidden_units: Sequence[int], bias: bool = True, activation: Optional[str] = "nn.ReLU",
                 use_bn: bool = False, dropout_ratio=None, use_ln: bool = False) -> None:
        super().__init__()
        self.hidden_units = list(hidden_units)
        if use_bn and use_ln:
            raise ValueError("Could not use_bn and use_ln at the same time in Perceptron.")
        if dropout_ratio is None or (isinstance(dropout_ratio, (list, tuple)) and len(dropout_ratio) == 0):
            drops = [0.0] * len(self.hidden_units)
        elif isinstance(dropout_ratio, (list, tuple)):
            drops = [float(x) for x in dropout_ratio]
            if len(drops) == 1:
                drops = drops * len(self.hidden_units)
            if len(drops) != len(self.hidden_units):
                raise ValueError("length of dropout_ratio and hidden_units must be same")
        else:
            drops = [float(dropout_ratio)] * len(self.hidden_units)
        self._plain = bias and activation == "nn.ReLU" and not use_bn and not use_ln and not any(d > 0 for d in drops)
        layers: List[nn.Module] = []
        d = in_features
        for h, dr in zip(self.hidden_units, drops):
            layers.append(nn.Linear(d, h, bias=False if use_bn else bias))
            if use_bn:
                layers.append(nn.BatchNorm1d(h))
            if use_ln:
                layers.append(nn.LayerNorm(h))
            act = _activation(activation)
            if act is not None:
                layers.append(act)
            if dr > 0.0:
                layers.append(nn.Dropout(dr))
            d = h
        self.mlp = nn.Sequential(*layers)

    def output_dim(self) -> int:
        return self.hidden_units[-1]

    def linears(self) -> List[nn.Linear]:
        return [m for m in self.mlp if isinstance(m, nn.Linear)]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._plain and _FUSED_MLP2 and x.dim() == 2 and (x.is_cuda or _on_emulator()):
            from .dense import mlp2, mlp2_fits

            lin = self.linears()
            if mlp2_fits(x, lin):  # a small two-layer stack on data (the bottom MLP): one launch each way
                return mlp2(x, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)
        if self._plain and _FUSED_RELU and x.is_cuda and x.dim() == 2:
            # Linear + bias + ReLU as one hipBLASLt call (ReLU in the GEMM epilogue): same values,
            # one launch less per layer than Linear followed by ReLU
            for m in self.mlp:
                if isinstance(m, nn.Linear):
                    x = _LinearReluFn.apply(x, m.weight, m.bias)
            return x
        return self.mlp(x)


def head_loss(model, dense: torch.Tensor, sparse: torch.Tensor, labels: torch.Tensor, d: Optional[torch.Tensor] = None):
    """(mean BCE-with-logits loss, logits [B]) of a DLRM head (`dense_mlp`, `final_mlp`, `output_mlp`, `dim`,
    `arch_with_sparse` of `model`) on the dense features and the pooled sparse block; shared by DLRM and ShardedDLRM."""
    if d is None:  # (`d`: the bottom MLP's output when the caller already ran it, e.g. under the rows all-to-all)
        d = model.dense_mlp(dense)
    fused = _FUSED_TOP_LOSS and model.final_mlp._plain and (sparse.is_cuda or _on_emulator())
    if fused and _FUSED_IA_TOP and model.arch_with_sparse and sparse.shape[0] >= _FUSED_IA_TOP_MIN_B:
        from .dense import interaction_top_fits, interaction_top_loss, top_loss_fits

        lin = model.final_mlp.linears()
        # (top_loss_fits only looks at the layer widths and the device of its first argument)
        if interaction_top_fits(d, sparse, model.dim, lin[0]) and top_loss_fits(sparse, lin, model.output_mlp):
            return interaction_top_loss(d, sparse, model.dim, lin[0], lin[1], model.output_mlp, labels)
    allf = dot_interaction(d, sparse, model.dim, cat_dense=True, cat_sparse=model.arch_with_sparse)
    if fused:
        from .dense import top_loss, top_loss_fits

        lin = model.final_mlp.linears()
        if top_loss_fits(allf, lin, model.output_mlp):
            return top_loss(allf, lin[0], lin[1], model.output_mlp, labels)
    logits = model.output_mlp(model.final_mlp(allf)).squeeze(1)
    return bce_with_logits(logits, labels), logits.detach()


class DLRM(nn.Module):
    """DLRM with dot interaction (``dlrm{}`` block of examples/dlrm_criteo.config)."""

    def __init__(
        self,
        tables: Sequence[EmbeddingBagConfig],
        sparse_features: Sequence[str],
        dense_dim: int,
        dense_mlp: Sequence[int] = (64, 16),
        final_mlp: Sequence[int] = (64, 32),
        arch_with_sparse: bool = True,
        num_class: int = 1,
        device: Optional[torch.device] = None,
        sparse_optimizer: Optional[SparseOptimizerConfig] = None,
        row_layout: str = "interleaved",
    ) -> None:
        super().__init__()
        dims = {f: t.embedding_dim for t in tables for f in t.feature_names}
        dset = {dims[f] for f in sparse_features}
        if len(dset) != 1:
            raise ValueError(f"sparse group feature dims must be the same, but we find {dset}")
        self.dim = dset.pop()
        if dense_mlp[-1] != self.dim:
            raise ValueError("dense mlp last hidden_unit must be the same sparse feature dim")
        self.num_sparse = len(sparse_features)
        self.arch_with_sparse = arch_with_sparse
        self.ebc = EmbeddingBagCollection(
            tables, device=device, optimizer=sparse_optimizer, groups={"sparse": list(sparse_features)},
            row_layout=row_layout,
        )
        self.dense_mlp = MLP(dense_dim, dense_mlp)
        n = self.num_sparse + 1
        feat = n * (n - 1) // 2 + self.dim + (self.num_sparse * self.dim if arch_with_sparse else 0)
        self.final_mlp = MLP(feat, final_mlp)
        self.output_mlp = OutputLinear(self.final_mlp.output_dim(), num_class)
        if device is not None:
            self.dense_mlp.to(device)
            self.final_mlp.to(device)
            self.output_mlp.to(device)

    def dense_parameters(self):
        for m in (self.dense_mlp, self.final_mlp, self.output_mlp):
            yield from m.parameters()

    def predict_from_embeddings(self, dense: torch.Tensor, sparse: torch.Tensor) -> torch.Tensor:
        d = self.dense_mlp(dense)
        if (not torch.is_grad_enabled() and _FUSED_IA_TOP and self.final_mlp._plain and self.arch_with_sparse
                and (sparse.is_cuda or _on_emulator()) and sparse.shape[0] >= _FUSED_IA_TOP_MIN_B):
            from .dense import interaction_first_layer, interaction_top_fits

            lin = self.final_mlp.linears()
            if interaction_top_fits(d, sparse, self.dim, lin[0]):  # inference: the interaction row stays on chip
                y = interaction_first_layer(d, sparse, self.dim, lin[0])
                for l in lin[1:]:
                    y = torch.relu(l(y))
                return self.output_mlp(y).squeeze(1)
        allf = dot_interaction(d, sparse, self.dim, cat_dense=True, cat_sparse=self.arch_with_sparse)
        return self.output_mlp(self.final_mlp(allf)).squeeze(1)

    def forward(self, dense: torch.Tensor, sparse_features: KeyedJaggedTensor) -> torch.Tensor:
        """-> logits [B] (probs = sigmoid(logits), rank_model.py:142-146)."""
        sparse = self.ebc.forward_grouped(sparse_features)["sparse"]
        return self.predict_from_embeddings(dense, sparse)

    def loss_from_embeddings(self, dense: torch.Tensor, sparse: torch.Tensor, labels: torch.Tensor):
        """(mean BCE-with-logits loss, logits [B]) -- what `TrainWrapper.forward` computes for a one-label rank model
        (/root/reference/tzrec/models/model.py:271-297, rank_model.py:219-262).  When the top MLP is the two-layer ReLU
        stack + one-unit output of the DLRM config, everything behind its first GEMM -- second layer, logit, loss and
        their whole backward -- is one launch (`dense.top_loss`); otherwise logits and loss the plain way."""
        return head_loss(self, dense, sparse, labels)

    def forward_loss(self, dense: torch.Tensor, sparse_features: KeyedJaggedTensor, labels: torch.Tensor):
        sparse = self.ebc.forward_grouped(sparse_features)["sparse"]
        return self.loss_from_embeddings(dense, sparse, labels)


class DeepFM(nn.Module):
    """DeepFM with groups wide / fm / deep (examples/deepfm_criteo.config)."""

    def __init__(
        self,
        tables: Sequence[EmbeddingBagConfig],
        groups: Dict[str, List[str]],
        dense_dim: int,
        fm_dim: int,
        deep_mlp: Sequence[int] = (512, 256, 128),
        final_mlp: Optional[Sequence[int]] = (64,),
        num_class: int = 1,
        device: Optional[torch.device] = None,
        sparse_optimizer: Optional[SparseOptimizerConfig] = None,
    ) -> None:
        super().__init__()
        self.ebc = EmbeddingBagCollection(tables, device=device, optimizer=sparse_optimizer, groups=groups)
        self.has_fm_group = "fm" in groups
        self.fm_dim = fm_dim
        dims = self.ebc._out_dim
        deep_in = sum(dims[k] for k in groups["deep"]) + dense_dim
        self.deep_mlp = MLP(deep_in, deep_mlp)
        self.fm = FactorizationMachine()
        final_dim = self.deep_mlp.output_dim()
        self.final_mlp = None
        if final_mlp:
            self.final_mlp = MLP(1 + fm_dim + final_dim, final_mlp)
            final_dim = self.final_mlp.output_dim()
        self.output_mlp = OutputLinear(final_dim, num_class)
        if device is not None:
            self.to(device)

    def dense_parameters(self):
        mods = [self.deep_mlp, self.output_mlp] + ([self.final_mlp] if self.final_mlp is not None else [])
        for m in mods:
            yield from m.parameters()

    def forward(self, dense: torch.Tensor, sparse_features: KeyedJaggedTensor) -> torch.Tensor:
        g = self.ebc.forward_grouped(sparse_features)
        y_wide = g["wide"].sum(dim=1, keepdim=True)
        # dense (raw) features are columns of the deep group, before the embeddings
        # (/root/reference/examples/deepfm_criteo.config deep group lists int_* first)
        deep = torch.cat([dense, g["deep"]], dim=1)
        y_deep = self.deep_mlp(deep)
        fm_in = g["fm"] if self.has_fm_group else g["deep"]
        y_fm = self.fm(fm_in.reshape(fm_in.shape[0], -1, self.fm_dim))
        if self.final_mlp is not None:
            y = self.output_mlp(self.final_mlp(torch.cat([y_wide, y_fm, y_deep], dim=1)))
        else:
            y = y_wide + y_fm.sum(dim=1, keepdim=True) + self.output_mlp(y_deep)
        return y.squeeze(1)


def bce_with_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """BCEWithLogitsLoss(reduction="mean") (rank_model.py:190-191,233-240): loss and d(loss)/d(logits)
    from one launch of `tzr_bce_logits` (torcheasyrec_amd/dense.py) instead of ~12 torch kernels."""
    from .dense import bce_with_logits as fused

    return fused(logits, labels)
