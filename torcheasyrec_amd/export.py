"""Quantized export of embedding tables.

Same function names, arguments and error behaviour as the reference's numpy encoder
(/root/reference/tzrec/utils/quant_util.py:18-196, called from tzrec/utils/export_util.py:2353 and
the delta-embedding dump), over `tzr_quantize_rows_q8f16` / `tzr_dequantize_rows_q8f16`: the table
stays in HBM, the encoded rows come back as a uint8 tensor on the same device, byte-for-byte what
the reference writes.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import _lib

DISTRIBUTED_SPARSE_QUANT_SCALE_OFFSET_BYTES = 4
DISTRIBUTED_SPARSE_SUPPORTED_QUANT_FORMATS: List[str] = ["QUint8RowwiseF16"]
_CHUNK_ROWS = 64 * 1024  # the reference validates in chunks of this many rows (quant_util.py:20-21,66-70)
_CHUNK_BYTES = 64 * 1024 * 1024
_NONE = (1 << 63) - 1


def _quantize_quint8_rowwise_f16(values: torch.Tensor, emb_dim: int, emb_name: str) -> torch.Tensor:
    if not isinstance(values, torch.Tensor):
        raise TypeError("values must be a torch.Tensor in HIP memory (the table itself, not a host copy)")
    if values.dim() != 2 or values.shape[1] != emb_dim:
        raise ValueError(f"Expected a 2D sparse embedding tensor with dim={emb_dim}, got shape={list(values.shape)}")
    if values.dtype not in (torch.float32, torch.float16):
        raise ValueError(f"embedding '{emb_name}': float32 or float16 rows expected, got {values.dtype}")
    row_bytes = emb_dim + DISTRIBUTED_SPARSE_QUANT_SCALE_OFFSET_BYTES
    if row_bytes % 2 != 0:
        raise ValueError(f"Distributed sparse quant export failed for embedding '{emb_name}': QUint8RowwiseF16 stores "
                         f"[uint8 values][float16 scale][float16 offset], row_bytes = {emb_dim} + 4 = {row_bytes} must be even")
    if emb_dim % 4 != 0:
        raise ValueError(f"embedding '{emb_name}': this library stores tables with embedding_dim % 4 == 0, got {emb_dim}")
    if values.stride(1) != 1:
        values = values.contiguous()
    dev = values.device
    rows = values.shape[0]
    out = torch.empty(rows, row_bytes, dtype=torch.uint8, device=dev)
    if rows == 0:
        return out
    bad = torch.empty(3, dtype=torch.int64, device=dev)
    dt = _lib.DT_F16 if values.dtype == torch.float16 else _lib.DT_F32
    _lib.check(_lib.lib().tzr_quantize_rows_q8f16(_lib.ptr(values), dt, values.stride(0), rows, emb_dim, _lib.ptr(out),
                                                  _lib.ptr(bad), _lib.stream_ptr(dev)), "tzr_quantize_rows_q8f16")
    nonfinite, offset, scale = (int(x) for x in bad.cpu().tolist())  # export is offline: one host sync
    pre = f"Distributed sparse quant export failed for embedding '{emb_name}': "
    if nonfinite != _NONE:
        raise ValueError(pre + "source values must all be finite")
    if offset != _NONE or scale != _NONE:
        # the reference walks chunks of rows and checks the offset before the scale inside a chunk
        chunk = min(_CHUNK_ROWS, max(1, _CHUNK_BYTES // max(emb_dim * 4, 1)))
        if offset != _NONE and (scale == _NONE or offset // chunk <= scale // chunk):
            v = float(values[offset].float().min())
            raise ValueError(pre + f"row {offset} offset {v} is outside the finite float16 range")
        r = values[scale].float()
        off16 = float(r.min().to(torch.float16).float())
        raise ValueError(pre + f"row {scale} scale {(float(r.max()) - off16) / 255.0} is outside the finite float16 range")
    return out


def distributed_quantize_embeddings(values: torch.Tensor, emb_dim: int, emb_name: str, quant_format: str) -> torch.Tensor:
    """Quantize embedding rows for the distributed export -> uint8 [rows, emb_dim + 4] on the device."""
    if quant_format == DISTRIBUTED_SPARSE_SUPPORTED_QUANT_FORMATS[0]:
        return _quantize_quint8_rowwise_f16(values, emb_dim, emb_name)
    raise ValueError(f"Unsupported distributed sparse quant format: {quant_format}; "
                     f"supported formats: {DISTRIBUTED_SPARSE_SUPPORTED_QUANT_FORMATS}")


def dequantize_quint8_rowwise_f16(rows: torch.Tensor, emb_dim: int) -> torch.Tensor:
    """uint8 [n, emb_dim + 4] QUint8RowwiseF16 rows -> float32 [n, emb_dim] (value * scale + offset)."""
    row_bytes = emb_dim + DISTRIBUTED_SPARSE_QUANT_SCALE_OFFSET_BYTES
    if rows.dim() != 2 or rows.shape[1] != row_bytes or rows.dtype != torch.uint8:
        raise ValueError(f"Expected a 2D QUint8RowwiseF16 array with row width={row_bytes} (embedding_dim={emb_dim}), "
                         f"got shape={list(rows.shape)}")
    rows = rows.contiguous()
    out = torch.empty(rows.shape[0], emb_dim, dtype=torch.float32, device=rows.device)
    _lib.check(_lib.lib().tzr_dequantize_rows_q8f16(_lib.ptr(rows), rows.shape[0], emb_dim, _lib.ptr(out), out.stride(0) if rows.shape[0] else emb_dim,
                                                    _lib.stream_ptr(rows.device)), "tzr_dequantize_rows_q8f16")
    return out


def quantize_tables(collection, quant_format: str = "QUint8RowwiseF16") -> Dict[str, torch.Tensor]:
    """{table: encoded rows} for every table of an EmbeddingBagCollection / EmbeddingCollection (the
    rows this rank holds when the collection is sharded)."""
    return {n: distributed_quantize_embeddings(w.detach(), w.shape[1], n, quant_format) for n, w in collection.table_weights().items()}
