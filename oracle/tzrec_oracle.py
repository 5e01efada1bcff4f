"""CPU oracle for the tzrec sharded-embedding hot path.  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker (or as the timed CPU baseline) -- never as a compute path of
``torcheasyrec_amd``.

What it restates.  The reference (``/root/reference``, alibaba/TorchEasyRec) is pure Python; the
arithmetic of this path lives in un-vendored wheels pinned in ``requirements/runtime.txt``:
``torchrec==1.7.0`` and ``fbgemm-gpu==1.7.0`` (absent here, no network).  This file restates their
*published* semantics and anchors them on the reference's own call sites:

  index stage (bit-exact)      tzrec/features/feature.py:80-166, tzrec/datasets/data_parser.py:526-594
  pooled lookup + regroup      tzrec/modules/embedding.py:909-978 -> torchrec EBC [upstream]
  fused sparse optimizer       tzrec/optim/optimizer_builder.py:30-97, protos/optimizer.proto:76-139
  dot interaction / FM         tzrec/modules/interaction.py:57-91, tzrec/modules/fm.py:17-42
  DLRM / DeepFM glue, loss     tzrec/models/dlrm.py:101-135, deepfm.py:72-108, rank_model.py:133-262

Pinning status.  The index stage is pinned against the reference's golden vectors
(tzrec/features/id_feature_test.py:41-69,169-188; tzrec/datasets/data_parser_test.py:36-154), see
tests/golden/reference_index_vectors.json and tests/test_oracle_golden.py.  Every floating-point
stage at the torchrec/fbgemm boundary is **parity unpinned**: the reference's tests assert shapes only
(SURVEY.md section 4 / 8c), so this oracle *defines* the expected values under these assumptions:
eps = 1e-8 (fbgemm default), duplicate rows of a step are summed (in lookup order) before ONE update,
empty bag -> 0, mean divides by bag length, Adagrad state starts at ``initial_accumulator_value`` (0),
gradient_clipping clamps the summed per-row gradient elementwise to +-max_gradient.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as tF

# --------------------------------------------------------------------------------------------
# index stage
# --------------------------------------------------------------------------------------------


def parse_sparse_feature(
    column: Sequence,
    default_value: Optional[List[int]] = None,
    multival_sep: str = "\x03",
    weighted: bool = False,
) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """FG-encoded sparse column -> (values int64, lengths int32, weights f32|None).

    Follows tzrec/features/feature.py:80-166 (`_parse_fg_encoded_sparse_feature_impl`):
    strings are split on ``multival_sep``; empty string and null become the default id list if one
    is configured, else a bag of length 0; integer columns drop nulls (length 0) or take
    ``default_value[0]``; ``weighted`` strings are ``id:weight`` pairs; dict rows are id->weight maps.
    """
    values: List[int] = []
    lengths: List[int] = []
    weights: List[float] = []
    has_w = weighted
    for x in column:
        ids: Optional[List[int]]
        ws: Optional[List[float]] = None
        if x is None:
            ids = None
        elif isinstance(x, str):
            if x == "":
                ids = None
            else:
                toks = x.split(multival_sep)
                if weighted:
                    ids, ws = [], []
                    for t in toks:
                        k, w = t.split(":")
                        ids.append(int(k))
                        ws.append(float(w))
                else:
                    ids = [int(t) for t in toks]
        elif isinstance(x, dict):
            has_w = True
            ids = [int(k) for k in x.keys()]
            ws = [float(v) for v in x.values()]
        elif isinstance(x, (list, tuple, np.ndarray)):
            ids = [int(t) for t in x]
            if len(ids) == 0 and default_value is not None:
                ids = None
        else:  # scalar integer
            ids = [int(x)]
        if ids is None:
            if default_value is not None:
                ids = list(default_value)
                ws = [1.0] * len(ids)
            else:
                ids = []
                ws = []
        if has_w and ws is None:
            ws = [1.0] * len(ids)
        values.extend(ids)
        lengths.append(len(ids))
        if has_w:
            weights.extend(ws or [])
    return (
        np.asarray(values, dtype=np.int64),
        np.asarray(lengths, dtype=np.int32),
        np.asarray(weights, dtype=np.float32) if has_w else None,
    )


def to_kjt(
    keys: Sequence[str],
    per_key_values: Sequence[np.ndarray],
    per_key_lengths: Sequence[np.ndarray],
    per_key_weights: Optional[Sequence[Optional[np.ndarray]]] = None,
) -> Dict[str, object]:
    """Per-feature (values, lengths) -> the KeyedJaggedTensor fields tzrec builds.

    Follows tzrec/datasets/data_parser.py:551-585: concatenation key-major; if ANY key of the data
    group is weighted every key gets weights (1.0 where it had none); stride = batch size.
    """
    any_w = per_key_weights is not None and any(w is not None for w in per_key_weights)
    ws = None
    if any_w:
        ws = np.concatenate(
            [
                (w if w is not None else np.ones(len(v), np.float32))
                for v, w in zip(per_key_values, per_key_weights)
            ]
        ).astype(np.float32)
    return {
        "keys": list(keys),
        "values": np.concatenate(per_key_values).astype(np.int64),
        "lengths": np.concatenate(per_key_lengths),
        "weights": ws,
        "stride": int(len(per_key_lengths[0])),
        "length_per_key": [int(len(v)) for v in per_key_values],
    }


def lengths_to_offsets(lengths: np.ndarray) -> np.ndarray:
    """offsets = [0] + cumsum(lengths), int64 (torchrec KJT.offsets() [upstream])."""
    out = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths.astype(np.int64), out=out[1:])
    return out


def bounds_check(values, offsets, rows_per_key, B, clamp: bool):
    """Count ids outside [0, rows) and optionally clamp them to 0 (fbgemm bounds_check_indices
    WARNING mode [upstream]; mode selected at tzrec/utils/plan_util.py:580,594)."""
    values = values.copy()
    bad = 0
    for f, rows in enumerate(rows_per_key):
        s, e = offsets[f * B], offsets[(f + 1) * B]
        seg = values[s:e]
        m = (seg < 0) | (seg >= rows)
        bad += int(m.sum())
        if clamp:
            seg[m] = 0
    return values, bad


def kjt_permute(permute, lengths, values, weights, B):
    """fbgemm permute_2D_sparse_data semantics [upstream]: output key t = input key permute[t]."""
    F = len(lengths) // B
    off = lengths_to_offsets(lengths)
    out_l, out_v, out_w = [], [], []
    for p in permute:
        assert 0 <= p < F
        out_l.append(lengths[p * B : (p + 1) * B])
        s, e = off[p * B], off[(p + 1) * B]
        out_v.append(values[s:e])
        if weights is not None:
            out_w.append(weights[s:e])
    cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)  # noqa: E731
    return (
        cat(out_l, lengths.dtype),
        cat(out_v, np.int64),
        cat(out_w, np.float32) if weights is not None else None,
    )


def splitmix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on int64 ids (two's complement), as the kernels hash raw ids."""
    with np.errstate(over="ignore"):
        z = x.astype(np.int64).view(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def block_bucketize(block_sizes, lengths, values, weights, B, W):
    """fbgemm block_bucketize_sparse_features semantics [upstream] for row-wise sharding.

    RW geometry: rank r owns rows [r*block, (r+1)*block) with block = ceil(rows/W)
    (tzrec/utils/plan_util.py:1049-1060 -> torchrec calculate_shard_sizes_and_offsets).
    Returns (new_lengths[W*F*B], new_values, new_weights, unbucketize_permute); output is
    rank-major, then key, then sample; ids keep their relative order inside a bag.
    """
    F = len(lengths) // B
    off = lengths_to_offsets(lengths)
    new_lengths = np.zeros(W * F * B, dtype=lengths.dtype)
    dest = np.empty(len(values), dtype=np.int64)
    for f in range(F):
        s, e = off[f * B], off[(f + 1) * B]
        if block_sizes[f] == 0:  # hash routing of raw ids (ZCH tables): splitmix64(id) mod W, id unchanged
            dest[s:e] = (splitmix64(values[s:e]) % np.uint64(W)).astype(np.int64)
        else:
            dest[s:e] = np.minimum(values[s:e] // block_sizes[f], W - 1)
    bag_of = np.repeat(np.arange(F * B, dtype=np.int64), lengths.astype(np.int64))
    slot = dest * (F * B) + bag_of  # output bag index (r*F + f)*B + b
    np.add.at(new_lengths, slot, 1)
    order = np.argsort(slot, kind="stable")
    f_of = bag_of // B
    local = values - dest * np.asarray(block_sizes, dtype=np.int64)[f_of]
    new_values = local[order]
    new_weights = weights[order] if weights is not None else None
    unbucketize = np.empty(len(values), dtype=np.int64)
    unbucketize[order] = np.arange(len(values), dtype=np.int64)
    return new_lengths, new_values, new_weights, unbucketize


# --------------------------------------------------------------------------------------------
# pooled lookup
# --------------------------------------------------------------------------------------------


@dataclass
class TableSpec:
    """What tzrec derives per table (tzrec/features/feature.py:611-636, embedding.py:692-884)."""

    name: str
    rows: int
    dim: int
    pooling: str = "sum"  # "sum" | "mean"


def pooled_lookup(
    weights_per_key: Sequence[torch.Tensor],
    poolings: Sequence[str],
    values: torch.Tensor,
    lengths: torch.Tensor,
    B: int,
    per_sample_weights: Optional[torch.Tensor] = None,
) -> List[torch.Tensor]:
    """Per KJT key: pooled [B, D] block (torchrec EBC unsharded semantics, SURVEY.md a7: one
    nn.EmbeddingBag(mode, include_last_offset=True) per table; empty bag -> 0; fp32 out).

    ``weights_per_key[f]`` is the table tensor key f reads (keys sharing a table pass the same
    tensor, so autograd sums their gradients).  Mean pooling with per-sample weights is not
    defined by ATen; it is restated as sum(w_i * row_i) / len(bag) (fbgemm TBE behaviour).
    """
    off = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lengths.to(torch.int64), 0)
    outs = []
    for f, (w, mode) in enumerate(zip(weights_per_key, poolings)):
        lo, hi = int(off[f * B]), int(off[(f + 1) * B])
        ids = values[lo:hi]
        o = off[f * B : (f + 1) * B + 1] - lo
        psw = per_sample_weights[lo:hi] if per_sample_weights is not None else None
        if mode == "mean" and psw is not None:
            s = tF.embedding_bag(ids, w, o, mode="sum", per_sample_weights=psw, include_last_offset=True)
            n = lengths[f * B : (f + 1) * B].to(torch.float32).clamp(min=1.0)
            outs.append(s / n[:, None])
        else:
            outs.append(
                tF.embedding_bag(ids, w, o, mode=mode, per_sample_weights=psw, include_last_offset=True)
            )
    return [o.float() for o in outs]


def regroup(
    blocks: Dict[str, torch.Tensor], groups: Dict[str, List[str]]
) -> Dict[str, torch.Tensor]:
    """KeyedTensor.regroup_as_dict: column-concat of a group's feature blocks in feature_names order
    (tzrec/modules/embedding.py:848-850,972-976); shared features are copied into every group."""
    return {g: torch.cat([blocks[n] for n in names], dim=1) for g, names in groups.items()}


@dataclass
class SparseOptim:
    """tzrec/optim/optimizer_builder.py:30-97 -> torchrec.optim Adagrad / RowWiseAdagrad / SGD."""

    kind: str = "adagrad"  # "sgd" | "adagrad" | "rowwise_adagrad"
    lr: float = 0.001
    eps: float = 1e-8
    weight_decay: float = 0.0
    weight_decay_mode: str = "none"  # rowwise adagrad: "none" | "l2" | "decouple"
    gradient_clipping: bool = False
    max_gradient: float = 1.0
    initial_accumulator_value: float = 0.0


def sparse_update(
    w: np.ndarray,
    m: Optional[np.ndarray],
    ids: np.ndarray,
    grads: np.ndarray,
    opt: SparseOptim,
) -> None:
    """Exact fused sparse update of ONE table, in place (fbgemm *_exact kernels [upstream]).

    ``ids[i]`` / ``grads[i, :]`` = row and dL/d(row contribution) of lookup i, in lookup order.
    Duplicates are summed sequentially in fp32 in that order (np.add.at), then each distinct row
    is updated once.
    """
    if len(ids) == 0:
        return
    uniq, inv, counts = np.unique(ids, return_inverse=True, return_counts=True)
    # stable sort keeps the lookup order inside each row's duplicates; reduceat then adds them
    # sequentially in fp32 (same result as np.add.at, ~50x faster)
    order = np.argsort(inv, kind="stable")
    starts = np.zeros(len(uniq), dtype=np.int64)
    np.cumsum(counts[:-1], out=starts[1:])
    g = np.add.reduceat(np.ascontiguousarray(grads, dtype=np.float32)[order], starts, axis=0)
    if opt.gradient_clipping:
        np.clip(g, -opt.max_gradient, opt.max_gradient, out=g)
    lr = np.float32(opt.lr)
    eps = np.float32(opt.eps)
    wd = np.float32(opt.weight_decay)
    rows = w[uniq].astype(np.float32)
    if opt.kind == "sgd":
        w[uniq] = rows - lr * g
    elif opt.kind == "adagrad":
        mm = m[uniq] + g * g
        m[uniq] = mm
        w[uniq] = rows - lr * g / (np.sqrt(mm) + eps)
    elif opt.kind == "rowwise_adagrad":
        gl = g + wd * rows if opt.weight_decay_mode == "l2" else g
        mm = m[uniq] + (gl * gl).mean(axis=1, dtype=np.float32)
        m[uniq] = mm
        mult = lr / (np.sqrt(mm) + eps)
        if opt.weight_decay_mode == "l2":
            corr = np.float32(1.0) - mult * wd
        elif opt.weight_decay_mode == "decouple":
            corr = np.full_like(mult, np.float32(1.0) - lr * wd)
        else:
            corr = np.ones_like(mult)
        w[uniq] = corr[:, None] * rows - mult[:, None] * g
    else:
        raise ValueError(opt.kind)


def lookup_grads(
    grad_blocks: Sequence[np.ndarray],
    lengths: np.ndarray,
    B: int,
    poolings: Sequence[str],
    per_sample_weights: Optional[np.ndarray] = None,
) -> np.ndarray:
    """dL/d(row contribution) per lookup, [N, D], from the per-key pooled-output gradients.

    Lookup i of bag (f, b) receives grad_blocks[f][b] * (weight_i) / (len(bag) if mean).
    All keys must share D.
    """
    F = len(grad_blocks)
    L = lengths.astype(np.int64)
    bag_of = np.repeat(np.arange(F * B, dtype=np.int64), L)
    g = np.concatenate(grad_blocks, axis=0).astype(np.float32)[bag_of]
    scale = np.ones(len(bag_of), dtype=np.float32)
    if per_sample_weights is not None:
        scale = scale * per_sample_weights.astype(np.float32)
    for f, mode in enumerate(poolings):
        if mode == "mean":
            sel = (bag_of // B) == f
            scale[sel] = scale[sel] / np.maximum(L[bag_of[sel]], 1).astype(np.float32)
    return g * scale[:, None]


# --------------------------------------------------------------------------------------------
# feature interaction (pure torch in the reference; restated, not imported)
# --------------------------------------------------------------------------------------------


def dot_interaction(x: torch.Tensor) -> torch.Tensor:
    """tzrec/modules/interaction.py:80-91: bmm(X, X^T), strict upper triangle, row-major (i<j)."""
    n = x.shape[1]
    z = torch.bmm(x, x.transpose(1, 2))
    iu = torch.triu_indices(n, n, offset=1)
    return z[:, iu[0], iu[1]]


def fm(x: torch.Tensor) -> torch.Tensor:
    """tzrec/modules/fm.py:37-41: 0.5 * ((sum_f x)^2 - sum_f x^2), [B,F,D] -> [B,D]."""
    s = x.sum(dim=1)
    return 0.5 * (s * s - (x * x).sum(dim=1))


def mlp(x: torch.Tensor, layers: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """tzrec/modules/mlp.py:58-83 with defaults: Linear + ReLU per hidden unit."""
    for w, b in layers:
        x = torch.relu(tF.linear(x, w, b))
    return x


def dlrm_forward(dense, sparse, p: Dict[str, object]) -> torch.Tensor:
    """tzrec/models/dlrm.py:101-135 (arch_with_sparse configurable) -> logits [B]."""
    B = sparse.shape[0]
    D = p["dim"]
    feat = sparse.reshape(B, -1, D)
    d = mlp(dense, p["dense_mlp"])
    feat = torch.cat([d.unsqueeze(1), feat], dim=1)
    allf = torch.cat([dot_interaction(feat), d], dim=-1)
    if p.get("arch_with_sparse", True):
        allf = torch.cat([allf, sparse], dim=-1)
    y = mlp(allf, p["final_mlp"])
    w, b = p["output"]
    return tF.linear(y, w, b).squeeze(1)


def deepfm_forward(wide, fm_in, deep, p: Dict[str, object]) -> torch.Tensor:
    """tzrec/models/deepfm.py:72-108 -> logits [B]."""
    B = fm_in.shape[0]
    y_wide = wide.sum(dim=1, keepdim=True)
    y_deep = mlp(deep, p["deep_mlp"])
    y_fm = fm(fm_in.reshape(B, -1, p["dim"]))
    w, b = p["output"]
    if p.get("final_mlp"):
        y = mlp(torch.cat([y_wide, y_fm, y_deep], dim=1), p["final_mlp"])
        y = tF.linear(y, w, b)
    else:
        y = y_wide + y_fm.sum(dim=1, keepdim=True) + tF.linear(y_deep, w, b)
    return y.squeeze(1)


def bce_with_logits(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """tzrec/models/rank_model.py:190-191,233-240: BCEWithLogitsLoss(reduction="mean") on float labels."""
    return tF.binary_cross_entropy_with_logits(logits, labels.float(), reduction="mean")


# --------------------------------------------------------------------------------------------
# sequence path
# --------------------------------------------------------------------------------------------


def jagged_to_padded_dense(values: torch.Tensor, lengths: torch.Tensor, max_len: int, pad: float = 0.0) -> torch.Tensor:
    """fbgemm jagged_to_padded_dense semantics [upstream] as used by JaggedTensor.to_padded_dense
    (tzrec/modules/embedding.py:1429,1480): [N, D] -> [B, max_len, D], truncating long sequences."""
    B, D = lengths.numel(), values.shape[1]
    out = torch.full((B, max_len, D), float(pad), dtype=values.dtype)
    s = 0
    rows = []
    for b in range(B):
        n = int(lengths[b])
        k = min(n, max_len)
        rows.append((b, s, k))
        s += n
    # built with differentiable ops so autograd gives the jagged gradient (zero on truncated rows)
    pieces = []
    for b, s0, k in rows:
        seg = values[s0:s0 + k]
        fill = torch.full((max_len - k, D), float(pad), dtype=values.dtype)
        pieces.append(torch.cat([seg, fill], dim=0))
    return torch.stack(pieces, dim=0) if pieces else out


def din_encoder(query, sequence, sequence_length, mlp_layers, linear):
    """tzrec/modules/sequence.py:101-128 (DINEncoder.forward), max_seq_length = 0."""
    L = sequence.shape[1]
    mask = torch.arange(L).unsqueeze(0) < sequence_length.unsqueeze(1)
    if query.shape[1] < sequence.shape[2]:
        query = tF.pad(query, (0, sequence.shape[2] - query.shape[1]))
    q = query.unsqueeze(1).expand(-1, L, -1)
    a = torch.cat([q, sequence, q - sequence, q * sequence], dim=-1)
    a = mlp(a, mlp_layers)
    a = tF.linear(a, linear[0], linear[1]).transpose(1, 2)
    scores = torch.where(mask.unsqueeze(1), a, torch.ones_like(a) * (-(2 ** 31) + 1))
    return torch.matmul(torch.softmax(scores, dim=-1), sequence).squeeze(1)
