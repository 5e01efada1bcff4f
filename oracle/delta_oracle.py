"""CPU restatement of the reference's delta-embedding tracker and dump rows.  TEST INFRASTRUCTURE ONLY:
imported by tests/, never by torcheasyrec_amd/.

Follows /root/reference/tzrec/utils/delta_embedding_dump.py:
  * `DeltaStore.append` / `get_unique`  -- `record_lookup` (:478-513) appends torch.cat of the ids of the
    features of one table per batch; `get_unique` (:565-609) returns torch.cat(...).unique() of the window
    and, with delete_on_read, forgets it;
  * `dump_rows` -- `_append_model_delta_rows` + `_lookup_embeddings` + `_append_table_chunk`
    (:962-1045, :1211-1275): per table ids.unique(sorted=True), rows = weight[ids], key_id = id + the shard's
    row offset, rows outside the local range raise ValueError.

Parity: UNPINNED against torchrec's DeltaStoreTrec (not installed here; the reference's tests for this
file need torchrec).  What it restates is only cat + unique, i.e. set semantics.
"""
from typing import Dict, List, Optional, Tuple

import numpy as np


class DeltaStore:
    def __init__(self, delete_on_read: bool = True) -> None:
        self.per_fqn: Dict[str, List[np.ndarray]] = {}
        self.delete_on_read = delete_on_read

    def append(self, fqn: str, ids: np.ndarray) -> None:
        self.per_fqn.setdefault(fqn, []).append(np.asarray(ids, dtype=np.int64).reshape(-1))

    def get_unique(self) -> Dict[str, np.ndarray]:
        out = {}
        for fqn, chunks in self.per_fqn.items():
            if not chunks:
                continue
            ids = np.unique(np.concatenate(chunks))
            if ids.size:
                out[fqn] = ids
        if self.delete_on_read:
            self.per_fqn = {}
        return out


def record_kjt(store: DeltaStore, feature_to_fqn: Dict[str, str], keys: List[str], values: np.ndarray, lengths: np.ndarray,
               stride: int) -> None:
    """record_lookup (:504-513): the ids of every feature of a table, concatenated, one append per table."""
    off = np.concatenate([[0], np.cumsum(lengths.astype(np.int64))])
    by_fqn: Dict[str, List[np.ndarray]] = {}
    for k, key in enumerate(keys):
        if key in feature_to_fqn:
            by_fqn.setdefault(feature_to_fqn[key], []).append(values[off[k * stride]:off[(k + 1) * stride]])
    for fqn, parts in by_fqn.items():
        store.append(fqn, np.concatenate(parts))


def dump_rows(ids: np.ndarray, weight: np.ndarray, row_offset: int = 0, fqn: str = "t") -> Tuple[np.ndarray, np.ndarray]:
    """(_lookup_embeddings :1024-1041) -> (float32 rows, global key ids)."""
    ids = np.unique(np.asarray(ids, dtype=np.int64))
    bad = (ids < 0) | (ids >= weight.shape[0])
    if bad.any():
        raise ValueError(f"Embedding table {fqn} was looked up with {int(bad.sum())} ids outside its local row range")
    return weight[ids].astype(np.float32), ids + row_offset


def bitmap_of(ids: np.ndarray, rows: int) -> np.ndarray:
    """The touched set as the uint32 word array the HIP tracker keeps (bit r & 31 of word r >> 5)."""
    words = np.zeros((rows + 31) // 32, dtype=np.uint32)
    ids = np.unique(ids[(ids >= 0) & (ids < rows)])
    np.bitwise_or.at(words, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return words
