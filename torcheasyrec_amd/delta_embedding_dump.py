"""Touched-row tracking for incremental embedding export: the device side of
`train_config.delta_embedding_dump_config` (/root/reference/tzrec/utils/delta_embedding_dump.py).

tzrec's `DeltaEmbeddingDumper` (parquet writer, dump cadence, config validation, file naming) is
control plane and stays tzrec's own class; what it needs from the embedding stack is a
`ModelDeltaTracker` with `get_unique(...)`, `clear`, `pause_tracking`, `fqn_to_feature_names`, and the
current rows of the ids it reports (reference :478-513, :565-609, :1043-1094).  That is what this module
provides -- INTEGRATION.md shows the two lines that bind it into tzrec's dumper -- plus
`published_rows()`, the (key ids, rows) pairs a dump writes, so the seam can be tested here.

What is different is where the touched ids live.  The reference inherits torchrec's
`ModelDeltaTracker` / `DeltaStoreTrec`: every lookup appends its id tensor, and `get_unique` runs
`torch.cat(...).unique()` over the window.  Here each tracked table owns a bitmap in HBM (one bit per
local row; `tzr_delta_mark` sets bits from the lookup's ids on the lookup's stream, `tzr_delta_count` /
`tzr_delta_collect` turn the bitmap into ascending ids at read time), so tracking memory is constant
(25.5 MB for all of DLRM-Criteo) and no sort ever runs.  The rows are read by `tzr_rows_gather`.

Zero-collision-hash tables publish RAW ids like the reference (:355-358, :515-550, :1043-1094): the ids
resident in the touched rows plus what the ZCH module reports per admission round (evicted, admitted,
looked up without a row), each with the row the table serves it from at read time (the shared fallback
row once an id holds none).  Not built: dynamicemb tables.
"""
from __future__ import annotations

from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib


@dataclass(frozen=True)
class _TableShardInfo:
    row_offset: int = 0
    local_rows: int = 0
    global_rows: int = 0
    global_cols: int = 0


@dataclass
class UniqueRows:
    ids: torch.Tensor
    states: Optional[torch.Tensor] = None


class _Site:
    """One lookup module whose lookups are recorded: the module that calls the tracker hook, the
    FQN prefix of its tables and, per table, where the local rows live."""

    def __init__(self, module_fqn: str, module: nn.Module, segment: str) -> None:
        self.module_fqn, self.module, self.segment = module_fqn, module, segment
        self.tables: Dict[str, Tuple[str, List[str], _TableShardInfo]] = {}  # table -> (fqn, features, shard)
        self.zch: Dict[str, object] = {}  # table -> ManagedCollisionModule (raw id per row)
        self.zch_wrapper = None  # the ManagedCollisionEmbeddingBagCollection holding their pending candidates

    def weights(self) -> Dict[str, torch.Tensor]:
        return self.module.table_weights()


def _clean_module_fqn(fqn: str) -> str:
    for wrapper in ("module.", "_dmp_wrapped_module."):
        while fqn.startswith(wrapper):
            fqn = fqn[len(wrapper):]
    return fqn


def _discover_sites(model: nn.Module) -> List[_Site]:
    """Walk the model like the reference walks for torchrec's sharded modules (:404-427).  A wrapper
    (sequence collection around its store, sharded collection around its local / replicated parts)
    is ONE site: its inner collections are not tracked a second time."""
    from .embedding import EmbeddingBagCollection
    from .sequence import EmbeddingCollection, ShardedEmbeddingCollection
    from .sharding import MixedShardedEmbeddingBagCollection, ShardedEmbeddingBagCollection
    from .zch import ManagedCollisionEmbeddingBagCollection, ShardedManagedCollisionEmbeddingBagCollection

    sites: List[_Site] = []
    consumed = set()

    def consume(m: nn.Module) -> None:
        for sub in m.modules():
            consumed.add(id(sub))

    for named_fqn, module in model.named_modules():
        if id(module) in consumed:
            continue
        fqn = _clean_module_fqn(named_fqn)
        if isinstance(module, (ManagedCollisionEmbeddingBagCollection, ShardedManagedCollisionEmbeddingBagCollection)):
            inner = module.ebc if isinstance(module, ManagedCollisionEmbeddingBagCollection) else module.sharded
            site = _Site(fqn, inner, "embedding_bags")
            site.zch_wrapper = module if isinstance(module, ManagedCollisionEmbeddingBagCollection) else module.mc
            site.zch = dict(site.zch_wrapper.modules_by_table)
            consume(module)
        elif isinstance(module, ShardedEmbeddingCollection):
            site = _Site(fqn, module.sharded, "embeddings")
            consume(module)
        elif isinstance(module, EmbeddingCollection):
            site = _Site(fqn, module, "embeddings")
            consume(module)
        elif isinstance(module, MixedShardedEmbeddingBagCollection):
            if module._cw:  # the reference refuses too (train.proto:169-171, delta_embedding_dump.py:253-266)
                name = sorted(module._cw)[0]
                k = len(module._cw[name])
                cols = next(c.embedding_dim for c in module._tables if c.name == name)
                raise ValueError("delta_embedding_dump_config does not support column-wise embedding sharding. Please use "
                                 f"table-wise, row-wise, or data-parallel sharding for table {name}. "
                                 f"local_cols={cols // k}, global_cols={cols}, column_offset=0.")
            # one site per exchange lane, all under the collection's own FQN (`<fqn>.embedding_bags.<table>`)
            for lane in module.lanes:
                lane_site = _Site(fqn, lane, "embedding_bags")
                for cfg in lane._global:
                    lo, n = lane.shard_of(cfg.name)
                    lane_site.tables[cfg.name] = (".".join(filter(None, (fqn, "embedding_bags", cfg.name))), list(cfg.feature_names),
                                                  _TableShardInfo(lo, n, cfg.num_embeddings, cfg.embedding_dim))
                sites.append(lane_site)
            consume(module)
            continue
        elif isinstance(module, (ShardedEmbeddingBagCollection, EmbeddingBagCollection)):
            site = _Site(fqn, module, "embedding_bags")
            consume(module)
        else:
            continue
        hook_mod = site.module
        if isinstance(hook_mod, ShardedEmbeddingBagCollection):
            for cfg in hook_mod._global:
                lo, n = hook_mod.shard_of(cfg.name)
                info = _TableShardInfo(lo, n, cfg.num_embeddings, cfg.embedding_dim)
                site.tables[cfg.name] = (".".join(filter(None, (fqn, site.segment, cfg.name))), list(cfg.feature_names), info)
        else:
            cfgs = hook_mod._store.embedding_bag_configs() if isinstance(hook_mod, EmbeddingCollection) else hook_mod.embedding_bag_configs()
            for cfg in cfgs:
                info = _TableShardInfo(0, cfg.num_embeddings, cfg.num_embeddings, cfg.embedding_dim)
                site.tables[cfg.name] = (".".join(filter(None, (fqn, site.segment, cfg.name))), list(cfg.feature_names), info)
        sites.append(site)
    return sites


class ModelDeltaTracker:
    """Touched embedding rows by owner-qualified table FQN (reference :352-641).

    Args as in the reference: `consumers` are independent readers of the id stream (each has its own
    bitmaps, so a read by one never hides ids from another), `delete_on_read` clears what a read
    returned, `auto_compact` is accepted and meaningless (a bitmap is always compact)."""

    DEFAULT_CONSUMER = "default"

    def __init__(self, model: nn.Module, consumers: Optional[List[str]] = None, delete_on_read: bool = True,
                 auto_compact: bool = False) -> None:
        self._consumers = list(consumers or [self.DEFAULT_CONSUMER])
        self._delete_on_read = delete_on_read
        self.curr_batch_idx = 0
        self.pause_depth = 0
        self.sites = _discover_sites(model)
        self.tracked_modules: Dict[str, nn.Module] = {}
        for s in self.sites:  # the lanes of a mixed-dim sharded collection share its FQN
            key, i = s.module_fqn, 0
            while key in self.tracked_modules:
                i += 1
                key = f"{s.module_fqn}#{i}"
            self.tracked_modules[key] = s.module
        self.fqn_to_feature_names: Dict[str, List[str]] = {}
        self.zch_modules: Dict[str, object] = {}
        self._shard_info: Dict[str, _TableShardInfo] = {}
        self._fqn_site: Dict[str, Tuple[_Site, str]] = {}
        self._bitmaps: Dict[str, Dict[str, torch.Tensor]] = {c: {} for c in self._consumers}
        self._seg_cache: Dict[tuple, List[torch.Tensor]] = {}
        self._device: Optional[torch.device] = None
        for site in self.sites:
            for table, (fqn, feats, info) in site.tables.items():
                if fqn in self.fqn_to_feature_names:
                    raise ValueError(f"Duplicate embedding table FQN: {fqn}")
                self.fqn_to_feature_names[fqn] = feats
                self._shard_info[fqn] = info
                self._fqn_site[fqn] = (site, table)
                if table in site.zch:
                    self.zch_modules[fqn] = site.zch[table]
                dev = site.module._device
                self._device = self._device or dev
                for c in self._consumers:
                    self._bitmaps[c][fqn] = torch.zeros((max(info.local_rows, 0) + 31) // 32, dtype=torch.int32, device=dev)
            site.module.register_post_lookup_tracker_fn(self._record_segments)
        self._site_of_module = {id(s.module): s for s in self.sites}
        # ZCH tables publish RAW ids (reference :355-358): rows touched -> the id resident in them, plus
        # what the ZCH module reports per admission round (evicted / admitted / looked up without a row)
        self._zch_fqn_by_mc_module = {id(m): fqn for fqn, m in self.zch_modules.items()}
        self._zch_events: Dict[str, Dict[str, List[torch.Tensor]]] = {c: {f: [] for f in self.zch_modules} for c in self._consumers}
        if self.zch_modules:
            from .zch import register_post_zch_event_tracker_fn

            for m in self.zch_modules.values():
                register_post_zch_event_tracker_fn(m, self.record_zch_event)
        self._oob = torch.zeros(1, dtype=torch.int64, device=self._device) if self._device is not None else None

    # -- recording -----------------------------------------------------------------------------
    @contextmanager
    def pause_tracking(self) -> Iterator[None]:
        """Stop recording lookups for non-training forward passes (reference :428-439)."""
        self.pause_depth += 1
        try:
            yield
        finally:
            self.pause_depth -= 1

    def _seg_arrays(self, site: _Site, segs: Tuple[Tuple[Optional[str], int], ...]) -> List[torch.Tensor]:
        ck = (id(site.module), segs)
        hit = self._seg_cache.get(ck)
        if hit is None:
            hit = []
            for c in self._consumers:
                arr = np.zeros(len(segs), dtype=_lib.DELTA_SEG_DT)
                for i, (table, key) in enumerate(segs):
                    arr[i]["key"] = key
                    if table is None:
                        continue
                    fqn, _, info = site.tables[table]
                    arr[i]["bitmap"] = self._bitmaps[c][fqn].data_ptr()
                    arr[i]["rows"] = info.local_rows
                hit.append(_lib.upload_struct(arr, site.module._device))
            self._seg_cache[ck] = hit
        return hit

    def _record_segments(self, emb_module: nn.Module, segs: Sequence[Tuple[Optional[str], int]], ids: torch.Tensor,
                         key_offsets: Optional[torch.Tensor], key_stride: int, uniform_len: int) -> None:
        """The hook the collections call after a lookup: segment i reads table segs[i][0] with the ids
        of key segment segs[i][1] (`tzr_delta_mark` for the addressing)."""
        if self.pause_depth > 0:
            return
        site = self._site_of_module.get(id(emb_module))
        if site is None:
            raise ValueError(f"Unrecognized embedding module for FQN delta tracking: {emb_module}")
        n = ids.numel()
        if n == 0 or not segs:
            return
        dev = site.module._device
        for d_segs in self._seg_arrays(site, tuple(segs)):
            _lib.check(_lib.lib().tzr_delta_mark(_lib.ptr(d_segs), len(segs), _lib.ptr(ids), _lib.ptr(key_offsets), key_stride,
                                                 uniform_len, n, _lib.ptr(self._oob), _lib.stream_ptr(dev)), "tzr_delta_mark")

    def record_lookup(self, kjt, states: Optional[torch.Tensor] = None, emb_module: Optional[nn.Module] = None,
                      raw_ids: Optional[torch.Tensor] = None) -> None:
        """Reference signature (:478-513): record the ids of `kjt` against the tables `emb_module`
        reads them with (ids are the module's LOCAL rows)."""
        if emb_module is None:
            raise ValueError("Embedding module is required for FQN delta tracking.")
        site = self._site_of_module.get(id(emb_module))
        if site is None:
            raise ValueError(f"Unrecognized embedding module for FQN delta tracking: {emb_module}")
        feature_to_table = {f: t for t, (_, feats, _) in site.tables.items() for f in feats}
        segs = tuple((feature_to_table[k], i) for i, k in enumerate(kjt.keys()))
        self._record_segments(emb_module, segs, kjt.values(), kjt.offsets(), kjt.stride(), 0)

    def record_zch_event(self, mc_module, evicted_raw_ids: torch.Tensor, admitted_raw_ids: torch.Tensor,
                         absent_raw_ids: Optional[torch.Tensor] = None) -> None:
        """Raw ids a ZCH table evicted / admitted this round (reference :515-550; recorded even while
        tracking is paused, :431-434) and, third, the ids looked up without a row."""
        fqn = self._zch_fqn_by_mc_module.get(id(mc_module))
        if fqn is None:
            raise ValueError(f"Unrecognized zch module for FQN delta tracking: {mc_module}")
        parts = [t for t in (admitted_raw_ids, evicted_raw_ids[evicted_raw_ids != _lib.ZCH_EMPTY]) if t.numel() > 0]
        if absent_raw_ids is not None and absent_raw_ids.numel() > 0 and self.pause_depth == 0:
            parts.append(absent_raw_ids)
        if parts:
            for c in self._consumers:
                self._zch_events[c][fqn].append(torch.cat(parts))

    # -- reading -------------------------------------------------------------------------------
    def get_unique(self, consumer: Optional[str] = None, top_percentage: Optional[float] = 1.0,
                   per_table_percentage=None, sorted_by_indices: Optional[bool] = True) -> Dict[str, UniqueRows]:
        """Unread touched rows per table FQN, ascending local row ids on the table's device; tables
        nothing touched are left out (reference :565-609)."""
        consumer = consumer or self._consumers[0]
        assert consumer in self._bitmaps, f"consumer {consumer} not present in {list(self._bitmaps)}"
        maps = self._bitmaps[consumer]
        fqns = [f for f in maps if self._shard_info[f].local_rows > 0]
        if not fqns:
            return {}
        L, dev = _lib.lib(), self._device
        totals = torch.zeros(len(fqns) + 1, dtype=torch.int64, device=dev)
        max_rows = max(self._shard_info[f].local_rows for f in fqns)
        ws = _lib.workspace(L.tzr_delta_collect_workspace(max_rows), dev)
        for i, f in enumerate(fqns):
            _lib.check(L.tzr_delta_count(_lib.ptr(maps[f]), self._shard_info[f].local_rows, _lib.ptr(totals[i:i + 1]),
                                         _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "tzr_delta_count")
        totals[-1:].copy_(self._oob)
        host = totals.cpu().tolist()  # the one host sync of a dump
        if host[-1] > 0:
            self._oob.zero_()
            raise ValueError(f"{host[-1]} looked-up ids were outside the local row range of their embedding table; "
                             "the feature's id space does not match the table it is embedded in.")
        out: Dict[str, UniqueRows] = {}
        for f, n in zip(fqns, host):
            ids = torch.empty(n, dtype=torch.int64, device=dev)
            if n:
                _lib.check(L.tzr_delta_collect(_lib.ptr(maps[f]), self._shard_info[f].local_rows, 0, 1 if self._delete_on_read else 0,
                                               _lib.ptr(ids), n, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), "tzr_delta_collect")
            if f in self.zch_modules:
                ids = self._zch_raw_ids(consumer, f, ids)
            if ids.numel():
                out[f] = UniqueRows(ids=ids, states=None)
        return out

    def _zch_raw_ids(self, consumer: str, fqn: str, rows: torch.Tensor) -> torch.Tensor:
        """Touched rows of a ZCH table -> the raw ids to publish: ids resident in those rows, ids the
        module admitted / evicted / saw without a row in the window, and the not-yet-coalesced
        candidates of the running round (those stay with the module and show up again next time)."""
        site, table = self._fqn_site[fqn]
        resident = self.zch_modules[fqn].row_ids[rows]
        parts = [resident[resident != _lib.ZCH_EMPTY]] + self._zch_events[consumer][fqn]
        if self.pause_depth == 0 and site.zch_wrapper is not None:
            parts.append(site.zch_wrapper.pending_candidates(table))
        if self._delete_on_read:
            self._zch_events[consumer][fqn] = []
        return torch.unique(torch.cat(parts))

    def get_unique_ids(self, consumer: Optional[str] = None) -> Dict[str, torch.Tensor]:
        return {fqn: rows.ids for fqn, rows in self.get_unique(consumer=consumer).items()}

    def published_rows(self, consumer: Optional[str] = None) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
        """{table FQN: (key ids int64[n], rows float32[n, D])} of everything touched since the last read:
        what one dump of the window writes.  Key ids are GLOBAL row ids (local row + the shard's row
        offset) or, for a ZCH table, raw ids, each with the row the table serves it from right now --
        its own row while it holds one, the shared fallback row once it does not."""
        out: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        for fqn, rows in self.get_unique(consumer=consumer).items():
            ids, weight = rows.ids, self.table_weight(fqn)
            zch = self.zch_modules.get(fqn)
            if zch is not None:
                out[fqn] = (ids, gather_rows(weight, zch.lookup_rows(ids)))
            else:
                out[fqn] = (ids + self._shard_info[fqn].row_offset, gather_rows(weight, ids))
        return out

    def step(self) -> None:
        self.curr_batch_idx += 1

    def trigger_compaction(self) -> None:
        """Nothing to do: the bitmap never holds a row twice."""

    def clear(self, consumer: Optional[str] = None) -> None:
        for c in ([consumer] if consumer is not None else self._consumers):
            assert c in self._bitmaps, f"consumer {c} not found in {list(self._bitmaps)}"
            for bm in self._bitmaps[c].values():
                bm.zero_()
            for f in self._zch_events[c]:
                self._zch_events[c][f] = []
        if self._oob is not None:
            self._oob.zero_()

    def shard_info(self, fqn: str) -> _TableShardInfo:
        return self._shard_info[fqn]

    def table_weight(self, fqn: str) -> torch.Tensor:
        site, table = self._fqn_site[fqn]
        return site.weights()[table]


def gather_rows(weight: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """float32 [n, D] copy of weight[ids] (fp32 or fp16 table rows, any row stride) -- `tzr_rows_gather`
    with one key segment."""
    dev, n, D = weight.device, ids.numel(), weight.shape[1]
    out = torch.empty(n, D, dtype=torch.float32, device=dev)
    if n == 0:
        return out
    tab = np.zeros(1, dtype=_lib.TABLE_DT)
    tab[0]["w"], tab[0]["rows"], tab[0]["dim"], tab[0]["w_stride"] = weight.data_ptr(), weight.shape[0], D, weight.stride(0)
    tab[0]["w_dtype"] = _lib.DT_F16 if weight.dtype == torch.float16 else _lib.DT_F32
    d_tab = _lib.upload_struct(tab, dev)
    key_table = torch.zeros(1, dtype=torch.int32, device=dev)
    key_start = torch.tensor([0, n], dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().tzr_rows_gather(_lib.ptr(d_tab), _lib.ptr(key_table), _lib.ptr(key_start), 1, _lib.ptr(ids), n,
                                          _lib.ptr(out), D, D, _lib.stream_ptr(dev)), "tzr_rows_gather")
    return out
