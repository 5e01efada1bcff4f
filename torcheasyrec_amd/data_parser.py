"""Host side of the index stage: arrow columns -> per-feature ids -> KeyedJaggedTensor / Batch.

SURVEY.md rows a1-a3.  In the reference this stage runs on CPU dataloader workers (pyarrow
compute) and hands the GPU a `Batch`; it is host code by nature, so it is host code here too:

  parse_sparse_column   <- _parse_fg_encoded_sparse_feature_impl
                           (/root/reference/tzrec/features/feature.py:80-166) and the FG_NONE branch
                           of BaseFeature._parse (:907-937)
  parse_sequence_column <- _parse_fg_encoded_sequence_sparse_feature_impl (:213-277)
  parse_dense_column    <- _parse_fg_encoded_dense_feature_impl (:169-210)
  DataParser.to_batch   <- DataParser._to_sparse_features / _to_dense_features
                           (/root/reference/tzrec/datasets/data_parser.py:502-594)

Every column type is first brought to one ragged form -- (row is present?, tokens per row, flat
tokens) -- and defaults / weights / casts are then plain numpy on that form.  Semantics pinned by
the reference's own test literals (tests/golden/reference_index_vectors.json):
  * a null or empty-string row has length 0, or holds `default` when one is configured;
  * null ints are dropped (length 0) unless a default is configured;
  * weighted strings are `id:weight` tokens, map columns give (key, item) = (id, weight); rows that
    take the default get weight 1.0;
  * if ANY key of a data group is weighted, every key of that group carries weights (1.0).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import torch

from .sparse import KeyedJaggedTensor, KeyedTensor


@dataclass
class SparseColumn:
    name: str
    values: np.ndarray  # int64 [sum lengths]
    lengths: np.ndarray  # int32 [B]          (sequence columns: ids per sequence STEP, see seq_lengths)
    weights: Optional[np.ndarray] = None  # float32 [sum lengths]
    seq_lengths: Optional[np.ndarray] = None  # int32 [B] steps per sample (sequence columns only)


@dataclass
class DenseColumn:
    name: str
    values: np.ndarray  # float32 [B, value_dim]


def _arrow(col) -> pa.Array:
    if isinstance(col, pa.ChunkedArray):
        return col.combine_chunks()
    if isinstance(col, pa.Array):
        return col
    col = list(col)
    if any(isinstance(x, dict) for x in col):  # python dicts mean id -> weight maps
        return pa.array([None if x is None else list(x.items()) for x in col], type=pa.map_(pa.string(), pa.float32()))
    return pa.array(col)


def apply_sample_mask(col, mask) -> pa.Array:
    """`use_mask` features during training: rows whose sample mask is set are nulled BEFORE parsing
    (BaseFeature.parse, tzrec/features/feature.py:862-887; the mask column is drawn per batch by the
    dataset, tzrec/datasets/dataset.py:356), so they take the default ids or an empty bag.  Map columns
    are left alone, as in the reference."""
    col = _arrow(col)
    if mask is None or pa.types.is_map(col.type):
        return col
    mask = pa.array(np.asarray(mask, dtype=bool)) if not isinstance(mask, (pa.Array, pa.ChunkedArray)) else mask
    return pc.if_else(mask, pa.nulls(len(col), col.type), col)


def _offsets_lengths(arr) -> np.ndarray:
    off = arr.offsets.to_numpy()
    return (off[1:] - off[:-1]).astype(np.int64)


class _Ragged:
    """present[B] (row carries data), counts[B] (tokens per present row, 0 otherwise), tokens (flat
    arrow array over the present rows only), aux (second flat array: map items / None)."""

    def __init__(self, present, counts, tokens, aux=None):
        self.present, self.counts, self.tokens, self.aux = present, counts, tokens, aux


def _ragged(col: pa.Array, sep: str, name: str) -> _Ragged:
    t = col.type
    n = len(col)
    notnull = np.ones(n, bool) if col.null_count == 0 else pc.is_valid(col).to_numpy(zero_copy_only=False)
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        nonempty = pc.fill_null(pc.not_equal(col, ""), False).to_numpy(zero_copy_only=False)
        present = notnull & nonempty
        kept = col.filter(pa.array(present))
        parts = pc.split_pattern(kept, sep)
        counts = np.zeros(n, np.int64)
        counts[present] = _offsets_lengths(parts)
        return _Ragged(present, counts, parts.values)
    if pa.types.is_map(t):
        kept = col.filter(pa.array(notnull))
        counts = np.zeros(n, np.int64)
        counts[notnull] = _offsets_lengths(kept)
        lo, hi = kept.offsets[0].as_py(), kept.offsets[len(kept)].as_py()
        return _Ragged(notnull, counts, kept.keys.slice(lo, hi - lo), kept.items.slice(lo, hi - lo))
    if pa.types.is_list(t) or pa.types.is_large_list(t):
        kept = col.filter(pa.array(notnull))
        counts = np.zeros(n, np.int64)
        counts[notnull] = _offsets_lengths(kept)
        return _Ragged(notnull & (counts > 0), counts, kept.flatten())
    if pa.types.is_integer(t) or pa.types.is_floating(t):
        return _Ragged(notnull, notnull.astype(np.int64), col.filter(pa.array(notnull)))
    raise ValueError(f"{name}: unsupported column type {t}")


def _interleave(present: np.ndarray, counts: np.ndarray, flat: np.ndarray, default: Optional[Sequence], dtype):
    """Rows without data take `default` (or nothing).  Returns (values, lengths)."""
    flat = np.asarray(flat, dtype=dtype)
    if default is None or bool(present.all()):
        return flat, np.where(present, counts, 0)
    d = np.asarray(list(default), dtype=dtype)
    lengths = np.where(present, counts, len(d))
    from_data = np.repeat(present, lengths)
    out = np.empty(int(lengths.sum()), dtype=dtype)
    out[from_data] = flat
    out[~from_data] = np.tile(d, int((~present).sum()))
    return out, lengths


def parse_sparse_column(name: str, col, multival_sep: str = chr(3), default_value: Optional[Sequence[int]] = None,
                        is_weighted: bool = False) -> SparseColumn:
    col = _arrow(col)
    if pa.types.is_floating(col.type):
        raise ValueError(f"{name} only support str|int|list<int>|map<int,double> dtype input, but get {col.type}.")
    if is_weighted and (pa.types.is_integer(col.type)):
        raise ValueError(f"{name}: an int column cannot be weighted")
    rg = _ragged(col, multival_sep, name)
    ids, w = rg.tokens, None
    if pa.types.is_map(col.type):
        w = rg.aux.cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False)
    elif is_weighted:
        if not (pa.types.is_string(ids.type) or pa.types.is_large_string(ids.type)):
            raise ValueError(f"{name}: weighted input must hold 'id:weight' strings")
        pair = pc.split_pattern(ids, ":")
        flat = pair.values
        ids = flat.take(pa.array(np.arange(0, len(flat), 2)))
        w = flat.take(pa.array(np.arange(1, len(flat), 2))).cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False)
    ids = ids.cast(pa.int64(), safe=False).to_numpy(zero_copy_only=False) if len(ids) else np.zeros(0, np.int64)
    values, lengths = _interleave(rg.present, rg.counts, ids, default_value, np.int64)
    weights = None
    if w is not None:
        weights, _ = _interleave(rg.present, rg.counts, w, None if default_value is None else [1.0] * len(default_value),
                                 np.float32)
    return SparseColumn(name, values, lengths.astype(np.int32), weights)


def parse_sequence_column(name: str, col, sequence_delim: str = ";", multival_sep: str = chr(3),
                          default_value: Optional[Sequence[int]] = None) -> SparseColumn:
    """Sequence of (possibly multi-valued) ids per sample: `seq_lengths[b]` steps, `lengths[s]` ids in
    step s, `values` all ids."""
    col = _arrow(col)
    t = col.type
    n = len(col)
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        rg = _ragged(col, sequence_delim, name)
        steps = rg.tokens  # one string per step
        ids = pc.split_pattern(steps, multival_sep)
        per_step = _offsets_lengths(ids)
        flat = ids.values.cast(pa.int64(), safe=False).to_numpy(zero_copy_only=False) if len(ids.values) else np.zeros(0, np.int64)
        seq = np.where(rg.present, rg.counts, 0)
        if default_value is not None and not rg.present.all():
            # a missing row is ONE step holding the default ids
            d = np.asarray(list(default_value), np.int64)
            seq = np.where(rg.present, rg.counts, 1)
            step_from_data = np.repeat(rg.present, seq)
            step_len = np.empty(int(seq.sum()), np.int64)
            step_len[step_from_data] = per_step
            step_len[~step_from_data] = len(d)
            id_from_data = np.repeat(step_from_data, step_len)
            out = np.empty(int(step_len.sum()), np.int64)
            out[id_from_data] = flat
            out[~id_from_data] = np.tile(d, int((~rg.present).sum()))
            flat, per_step = out, step_len
        return SparseColumn(name, flat, per_step.astype(np.int32), None, seq.astype(np.int32))
    if pa.types.is_list(t):
        nested = pa.types.is_list(t.value_type)
        rows = col.to_pylist()
        vals: List[int] = []
        step_len: List[int] = []
        seq = np.zeros(n, np.int32)
        for b, r in enumerate(rows):
            if (r is None or len(r) == 0) and default_value is not None:
                r = [list(default_value)] if nested else list(default_value)
            r = r or []
            seq[b] = len(r)
            for step in r:
                step = list(step) if nested else [step]
                vals.extend(int(x) for x in step)
                step_len.append(len(step))
        return SparseColumn(name, np.asarray(vals, np.int64), np.asarray(step_len, np.int32), None, seq)
    raise ValueError(f"{name}: unsupported sequence column type {t}")


def parse_dense_column(name: str, col, multival_sep: str = chr(3), default_value: Optional[Sequence[float]] = None
                       ) -> DenseColumn:
    col = _arrow(col)
    rg = _ragged(col, multival_sep, name)
    flat = rg.tokens.cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False) if len(rg.tokens) else np.zeros(0, np.float32)
    values, lengths = _interleave(rg.present, rg.counts, flat, default_value, np.float32)
    if len(lengths) and (lengths != lengths[0]).any():
        bad = int(np.nonzero(lengths != lengths[0])[0][0])
        raise ValueError(f"{name}: row {bad} has {int(lengths[bad])} values, row 0 has {int(lengths[0])} "
                         "(null dense values need a default)")
    dim = int(lengths[0]) if len(lengths) else 1
    return DenseColumn(name, values.reshape(len(lengths), dim))


def parse_label_column(name: str, col) -> torch.Tensor:
    """Label column -> tensor (tzrec/datasets/data_parser.py:226-262): float -> float32, int -> int64
    (list-typed labels of the generative models are outside this path)."""
    col = _arrow(col)
    if pa.types.is_floating(col.type):
        return torch.from_numpy(np.array(col.cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False)))
    if pa.types.is_integer(col.type):
        return torch.from_numpy(np.array(col.cast(pa.int64(), safe=False).to_numpy(zero_copy_only=False)))
    raise ValueError(f"label column [{name}] only support int | float dtype now.")


def parse_sample_weight_column(name: str, col) -> torch.Tensor:
    """tzrec/datasets/data_parser.py:264-272: sample weights must be a float column."""
    col = _arrow(col)
    if pa.types.is_floating(col.type):
        return torch.from_numpy(np.array(col.cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False)))
    raise ValueError(f"sample weight column [{name}] should be float dtype.")


@dataclass
class SequenceDenseColumn:
    name: str
    values: np.ndarray  # float32 [sum seq_lengths, value_dim]
    seq_lengths: np.ndarray  # int [B] steps per sample


def parse_sequence_dense_column(name: str, col, sequence_delim: str = ";", multival_sep: str = chr(3), value_dim: int = 1,
                                default_value: Optional[Sequence[float]] = None) -> SequenceDenseColumn:
    """Sequence of `value_dim`-wide float rows per sample
    (tzrec/features/feature.py:281-343, `_parse_fg_encoded_sequence_dense_feature_impl`): a string
    column holds `step;step;...` with `multival_sep` inside a step, list columns hold
    list<list<float>> (or list<float> when value_dim == 1).  A missing / empty row is ONE step
    holding the default."""
    col = _arrow(col)
    t = col.type
    if pa.types.is_string(t) or pa.types.is_large_string(t):
        rg = _ragged(col, sequence_delim, name)
        flat = pc.split_pattern(rg.tokens, multival_sep).values
        data = flat.cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False) if len(flat) else np.zeros(0, np.float32)
        steps = rg.counts
    elif pa.types.is_list(t) or pa.types.is_large_list(t):
        rg = _ragged(col, "", name)
        inner = rg.tokens
        if pa.types.is_list(inner.type) or pa.types.is_large_list(inner.type):
            inner = inner.flatten()
        data = inner.cast(pa.float32(), safe=False).to_numpy(zero_copy_only=False) if len(inner) else np.zeros(0, np.float32)
        steps = rg.counts
    else:
        raise ValueError(f"{name} only support str|list<float>|list<list<float>> dtype input, but get {t}.")
    data = np.asarray(data, np.float32).reshape(-1, value_dim)
    present = rg.present
    seq = np.where(present, steps, 0)
    if default_value is not None and not bool(present.all()):
        d = np.asarray(list(default_value), np.float32).reshape(1, -1)
        seq = np.where(present, steps, 1)
        from_data = np.repeat(present, seq)
        out = np.empty((int(seq.sum()), d.shape[1]), np.float32)
        out[from_data] = data
        out[~from_data] = d
        data = out
    return SequenceDenseColumn(name, data, seq)


class DataParser:
    """Feature columns of one data group -> tensors of a batch.

    `sparse_keys` / `dense_keys` fix the key order (the order of the feature configs, as the
    reference does); `sequence_keys` are flattened like the reference's multi-value sequences: the
    KJT length of a sample is the total number of ids over its steps."""

    def __init__(self, sparse_keys: Sequence[str], dense_keys: Sequence[str] = (), sequence_keys: Sequence[str] = (),
                 sequence_mulval_keys: Sequence[str] = ()):
        self.sparse_keys, self.dense_keys = list(sparse_keys), list(dense_keys)
        self.sequence_keys = set(sequence_keys)
        self.sequence_mulval_keys = [k for k in self.sparse_keys if k in set(sequence_mulval_keys)]

    def to_mulval_lengths(self, cols: Dict[str, SparseColumn]) -> Optional[KeyedJaggedTensor]:
        """Sequence features with value_dim != 1: KJT whose `values` are the ids per sequence step
        (key_lengths) and whose `lengths` are the steps per sample (tzrec/datasets/data_parser.py:554-593);
        `to_kjt` carries their ids flattened, one bag of all the sample's ids per sample."""
        if not self.sequence_mulval_keys:
            return None
        cs = [cols[k] for k in self.sequence_mulval_keys]
        for k, c in zip(self.sequence_mulval_keys, cs):
            if c.seq_lengths is None:
                raise ValueError(f"{k}: not a sequence column (parse it with parse_sequence_column)")
        return KeyedJaggedTensor(self.sequence_mulval_keys,
                                 torch.from_numpy(np.concatenate([c.lengths for c in cs]).astype(np.int64)),
                                 torch.from_numpy(np.concatenate([c.seq_lengths for c in cs]).astype(np.int32)))

    def to_kjt(self, cols: Dict[str, SparseColumn]) -> KeyedJaggedTensor:
        any_weighted = any(cols[k].weights is not None for k in self.sparse_keys)
        vals, lens, wts = [], [], []
        for k in self.sparse_keys:
            c = cols[k]
            ln = c.lengths
            if c.seq_lengths is not None:  # ids per sample = sum of ids over its steps
                ends = np.cumsum(c.seq_lengths)
                csum = np.concatenate([[0], np.cumsum(c.lengths)])
                ln = (csum[ends] - csum[ends - c.seq_lengths]).astype(np.int32)
            vals.append(c.values)
            lens.append(ln)
            if any_weighted:
                wts.append(c.weights if c.weights is not None else np.ones(len(c.values), np.float32))
        B = len(lens[0])
        for k, ln in zip(self.sparse_keys, lens):
            if len(ln) != B:
                raise ValueError(f"{k}: {len(ln)} rows, batch has {B}")
        return KeyedJaggedTensor(
            self.sparse_keys, torch.from_numpy(np.concatenate(vals).astype(np.int64)),
            torch.from_numpy(np.concatenate(lens).astype(np.int32)),
            torch.from_numpy(np.concatenate(wts).astype(np.float32)) if any_weighted else None)

    @staticmethod
    def to_sequence_dense(cols: Dict[str, "SequenceDenseColumn"]) -> Dict[str, object]:
        """{feature: JaggedTensor(values [steps, value_dim], lengths = steps per sample)} for
        `Batch.sequence_dense_features` (tzrec/datasets/data_parser.py:450-456)."""
        from .sequence import JaggedTensor

        return {k: JaggedTensor.from_lengths(torch.from_numpy(np.array(c.values, dtype=np.float32)),
                                             torch.from_numpy(np.asarray(c.seq_lengths).astype(np.int32)))
                for k, c in cols.items()}

    def to_keyed_tensor(self, cols: Dict[str, DenseColumn]) -> KeyedTensor:
        mats = [cols[k].values for k in self.dense_keys]
        return KeyedTensor(self.dense_keys, [m.shape[1] for m in mats],
                           torch.from_numpy(np.concatenate(mats, axis=1).astype(np.float32)))
