"""Delta-embedding dump: publish the rows touched since the last dump.

Mirror of the reference's `tzrec/utils/delta_embedding_dump.py` for this package's embedding
collections: same class names (`ModelDeltaTracker`, `DeltaEmbeddingDumper`), same config fields
(`DeltaEmbeddingDumpConfig`, /root/reference/tzrec/protos/train.proto:86-111), same cadence rules
(`maybe_dump` / `final_dump`, :812-872), same parquet schema, file naming and atomic write
(:76-99, :947-960, :1211-1313), same call sites in the train loop
(/root/reference/tzrec/main.py:449,517,547,611,900,928).

What is different is where the touched ids live.  The reference inherits torchrec's
`ModelDeltaTracker` / `DeltaStoreTrec`: every lookup appends its id tensor, and `get_unique` runs
`torch.cat(...).unique()` over the window (:478-513, :565-609).  Here each tracked table owns a bitmap
in HBM (one bit per local row; `tzr_delta_mark` sets bits from the lookup's ids on the lookup's
stream, `tzr_delta_count` / `tzr_delta_collect` turn the bitmap into ascending ids at dump time), so
tracking memory is constant (25.5 MB for all of DLRM-Criteo) and no sort ever runs.  The rows are
read by `tzr_rows_gather` and, for `quant_type: DELTA_EMBEDDING_QUANT_INT8`, encoded by
`tzr_quantize_rows_q8f16` on the device; only the finished bytes cross PCIe.

Zero-collision-hash tables publish RAW ids like the reference (:355-358, :515-550, :1043-1094): the ids
resident in the touched rows plus what the ZCH module reports per admission round (evicted, admitted,
looked up without a row), each with the row the table serves it from at dump time (the shared fallback
row once an id holds none).  Not built: the FeatureStore uploader (`feature_store_config`, a network
service) and dynamicemb tables.
"""
from __future__ import annotations

import os
import time
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from .export import (DISTRIBUTED_SPARSE_QUANT_SCALE_OFFSET_BYTES, DISTRIBUTED_SPARSE_SUPPORTED_QUANT_FORMATS,
                     distributed_quantize_embeddings)

_CONSUMER = "delta_embedding_dump"
QUANT_NONE = "DELTA_EMBEDDING_QUANT_NONE"
QUANT_INT8 = "DELTA_EMBEDDING_QUANT_INT8"


def _schema(quantized: bool):
    import pyarrow as pa

    return pa.schema([
        ("global_step", pa.int64()),
        ("rank", pa.int32()),
        ("world_size", pa.int32()),
        ("feature_name", pa.string()),
        ("table_fqn", pa.string()),
        ("key_id", pa.int64()),
        ("embedding", pa.list_(pa.uint8() if quantized else pa.float32())),
        ("source", pa.string()),
    ])


@dataclass
class DeltaEmbeddingDumpConfig:
    """train.proto:86-111.  Unset optional fields are None (`HasField` semantics)."""

    dump_interval_steps: Optional[int] = None  # proto default 1000
    output_dir: str = ""
    file_prefix: str = "delta_embedding"
    dump_interval_minutes: Optional[int] = None
    quant_type: str = QUANT_NONE
    feature_store_config: Optional[object] = None

    def HasField(self, name: str) -> bool:
        return getattr(self, name) is not None

    @property
    def interval_steps(self) -> int:
        return 1000 if self.dump_interval_steps is None else int(self.dump_interval_steps)


def delta_embedding_dump_config_from_msg(msg) -> DeltaEmbeddingDumpConfig:
    """`train_config { delta_embedding_dump_config { ... } }` of a parsed text-format config."""
    q = msg.one("quant_type", QUANT_NONE)
    if q not in (QUANT_NONE, QUANT_INT8):
        raise ValueError(f"delta_embedding_dump_config.quant_type: unknown value {q!r}")
    return DeltaEmbeddingDumpConfig(
        dump_interval_steps=msg.one("dump_interval_steps") if msg.has("dump_interval_steps") else None,
        output_dir=msg.one("output_dir", ""), file_prefix=msg.one("file_prefix", "delta_embedding"),
        dump_interval_minutes=msg.one("dump_interval_minutes") if msg.has("dump_interval_minutes") else None,
        quant_type=q, feature_store_config=msg.one("feature_store_config") if msg.has("feature_store_config") else None)


def validate_delta_embedding_dump_config(config: Optional[DeltaEmbeddingDumpConfig], device: torch.device) -> None:
    """Reference :128-155, same messages.  (The emulator library of the tests stands in for the GPU.)"""
    if config is None:
        return
    if torch.device(device).type != "cuda" and _lib.backend() != "emu":
        raise ValueError(f"delta_embedding_dump_config only supports CUDA training, but got device={device}.")
    if config.HasField("dump_interval_minutes"):
        if config.HasField("dump_interval_steps"):
            raise ValueError("delta_embedding_dump_config must configure only one of "
                             "dump_interval_steps and dump_interval_minutes.")
        if config.dump_interval_minutes <= 0:
            raise ValueError("delta_embedding_dump_config.dump_interval_minutes must be > 0.")
    elif config.interval_steps <= 0:
        raise ValueError("delta_embedding_dump_config.dump_interval_steps must be > 0.")
    if config.feature_store_config is not None:
        raise NotImplementedError("delta_embedding_dump_config.feature_store_config: the FeatureStore uploader is a "
                                  "network service outside this library; dump to output_dir and upload from there")


def _distributed_rank_world_size() -> Tuple[int, int]:
    rank = int(os.environ.get("RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
        world_size = torch.distributed.get_world_size()
    return rank, world_size


def _feature_name(feature_names: Iterable[str]) -> str:
    names = list(feature_names)
    return names[0] if len(names) == 1 else ",".join(names)


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


class DeltaEmbeddingDumper:
    """Dump touched embedding ids and their latest rows during training (reference :643-1313).

    Args:
        model: the model holding the embedding collections to track.
        config: DeltaEmbeddingDumpConfig.
        model_dir: base directory; `<model_dir>/delta_embedding_dump` is the default output location.
        device: training device."""

    def __init__(self, model: nn.Module, config: DeltaEmbeddingDumpConfig, model_dir: str, device: torch.device) -> None:
        validate_delta_embedding_dump_config(config, device)
        self._model, self._config = model, config
        self._quant_type = config.quant_type
        self._quantized = self._quant_type == QUANT_INT8
        self._schema = _schema(self._quantized)
        self._interval_steps: Optional[int] = None
        self._interval_secs: Optional[float] = None
        if config.HasField("dump_interval_minutes"):
            self._interval_secs = float(config.dump_interval_minutes * 60)
        else:
            self._interval_steps = config.interval_steps
        self._next_dump_time: Optional[float] = None
        self._last_dump_step: Optional[int] = None
        self._output_dir = config.output_dir or os.path.join(model_dir, "delta_embedding_dump")
        self._file_prefix = config.file_prefix or "delta_embedding"
        self._rank, self._world_size = _distributed_rank_world_size()
        os.makedirs(self._output_dir, exist_ok=True)
        self._tracker = ModelDeltaTracker(model, consumers=[_CONSUMER], delete_on_read=True, auto_compact=True)
        self._zch_modules = self._tracker.zch_modules
        if self._quantized:
            for fqn in self._tracker.fqn_to_feature_names:
                cols = self._tracker.shard_info(fqn).global_cols
                if cols % 2 != 0:
                    raise ValueError("delta_embedding_dump_config.quant_type=INT8 requires even "
                                     f"embedding_dim, but table '{fqn}' has emb_dim={cols}. QUint8RowwiseF16 format requires "
                                     f"row_bytes=emb_dim+{DISTRIBUTED_SPARSE_QUANT_SCALE_OFFSET_BYTES} to be even.")

    @property
    def tracker(self) -> ModelDeltaTracker:
        return self._tracker

    def clear(self) -> None:
        """Drop what was tracked so far, usually after restore-time dummy steps (reference :761-770)."""
        self._tracker.clear(_CONSUMER)

    @contextmanager
    def pause_tracking(self) -> Iterator[None]:
        with self._tracker.pause_tracking():
            yield

    def start(self) -> None:
        if self._interval_secs is not None:
            self._next_dump_time = time.monotonic() + self._interval_secs

    def close(self, raise_on_error: bool = True, drain: bool = True) -> None:
        """(the reference closes its FeatureStore uploader here)"""

    # -- cadence (reference :812-895) -------------------------------------------------------------
    def maybe_dump(self, global_step: int) -> None:
        if self._local_dump_decision(global_step):
            self.dump(global_step)
            self._last_dump_step = global_step
            if self._interval_secs is not None and self._next_dump_time is not None:
                now = time.monotonic()  # fixed-rate rescheduling; missed deadlines are skipped, not fired as a burst
                while self._next_dump_time <= now:
                    self._next_dump_time += self._interval_secs
        self._tracker.step()

    def _local_dump_decision(self, global_step: int) -> bool:
        if self._interval_steps is not None:
            return global_step > 0 and global_step % self._interval_steps == 0
        if self._interval_secs is not None and self._next_dump_time is not None:
            return time.monotonic() >= self._next_dump_time
        return False

    def final_dump(self, global_step: int) -> Optional[str]:
        """Flush the trailing partial interval at the end of training; boundary steps were already
        written by `maybe_dump` and are skipped (re-dumping would overwrite them with an empty shard)."""
        if global_step <= 0:
            return None
        global_step = self._sync_final_step(global_step)
        if self._interval_steps is not None and global_step % self._interval_steps == 0:
            return None
        if self._interval_secs is not None and global_step == self._last_dump_step:
            return None
        return self.dump(global_step)

    def _sync_final_step(self, global_step: int) -> int:
        """MAX over ranks, so every rank takes the same skip / dump decision into the same directory."""
        if self._world_size <= 1 or not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return global_step
        dev = self._tracker._device if torch.distributed.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(global_step, dtype=torch.long, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return int(t.item())

    # -- dump (reference :897-1045, :1211-1313) ---------------------------------------------------
    def dump(self, global_step: int) -> Optional[str]:
        """Write the tracked ids and their current rows to one parquet file; returns its path, or None
        when a single-process run had nothing to write."""
        global_step = int(global_step)
        if global_step <= 0:
            raise ValueError("delta embedding dump global_step must be > 0")
        chunks: list = []
        num_rows = 0
        for fqn, unique_rows in self._tracker.get_unique(_CONSUMER).items():
            ids = unique_rows.ids
            if ids.numel() == 0:
                continue
            embeddings, key_ids = self._lookup_embeddings(fqn, ids)
            feature_name = _feature_name(self._tracker.fqn_to_feature_names.get(fqn, []))
            num_rows += self._append_table_chunk(chunks, global_step, feature_name, fqn, key_ids, embeddings, "model_delta_tracker")
        output_path: Optional[str] = None
        if num_rows > 0 or self._world_size > 1:
            # multi-rank shard sets stay complete even for an empty rank
            output_path = self._output_path(global_step)
            self._write_table_chunks(chunks, output_path)
        return output_path

    def _output_path(self, global_step: int) -> str:
        if self._world_size == 1:
            return os.path.join(self._output_dir, f"{self._file_prefix}_step_{global_step}.parquet")
        step_dir = os.path.join(self._output_dir, f"step_{global_step}")
        os.makedirs(step_dir, exist_ok=True)
        return os.path.join(step_dir, f"{self._file_prefix}_step_{global_step}_rank_{self._rank}_of_{self._world_size}.parquet")

    def _lookup_embeddings(self, fqn: str, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        weight = self._tracker.table_weight(fqn)
        zch = self._zch_modules.get(fqn)
        if zch is not None:
            # `ids` are RAW ids: publish the row the table serves each of them from right now -- its own
            # row while it holds one, the shared fallback row once it does not (reference :1043-1094)
            return gather_rows(weight, zch.lookup_rows(ids)), ids
        return gather_rows(weight, ids), ids + self._tracker.shard_info(fqn).row_offset

    def _append_table_chunk(self, table_chunks: list, global_step: int, feature_name: str, table_fqn: str,
                            key_ids: torch.Tensor, embeddings: torch.Tensor, source: str) -> int:
        import pyarrow as pa

        if embeddings.dim() != 2:
            raise ValueError(f"delta embedding dump expects a 2-D embedding tensor, but got shape={tuple(embeddings.shape)}.")
        num_rows = int(key_ids.numel())
        if num_rows == 0:
            return 0
        if embeddings.size(0) != num_rows:
            raise ValueError("delta embedding dump key ids and embeddings row count mismatch: "
                             f"key_ids={num_rows}, embeddings={embeddings.size(0)}.")
        if self._quantized:
            try:  # encoded on the device; only the bytes travel
                embeddings = distributed_quantize_embeddings(embeddings, embeddings.size(1), feature_name,
                                                             DISTRIBUTED_SPARSE_SUPPORTED_QUANT_FORMATS[0])
            except ValueError as e:
                raise ValueError(f"Delta embedding dump INT8 quantization failed for feature '{feature_name}' "
                                 f"(table '{table_fqn}'): {e}. Disable delta dump quantization by setting "
                                 "delta_embedding_dump_config.quant_type to DELTA_EMBEDDING_QUANT_NONE.") from e
            value_type = pa.uint8()
        else:
            value_type = pa.float32()
        key_ids_cpu = key_ids.detach().cpu().to(torch.int64).contiguous()
        embeddings_cpu = embeddings.detach().cpu().contiguous()
        emb_dim = embeddings_cpu.size(1)
        offsets = np.arange(0, (num_rows + 1) * emb_dim, emb_dim, dtype=np.int32) if emb_dim else np.zeros(num_rows + 1, np.int32)
        values = pa.array(embeddings_cpu.reshape(-1).numpy(), type=value_type)
        table_chunks.append(pa.Table.from_arrays([
            pa.repeat(pa.scalar(global_step, pa.int64()), num_rows),
            pa.repeat(pa.scalar(self._rank, pa.int32()), num_rows),
            pa.repeat(pa.scalar(self._world_size, pa.int32()), num_rows),
            pa.repeat(pa.scalar(feature_name, pa.string()), num_rows),
            pa.repeat(pa.scalar(table_fqn, pa.string()), num_rows),
            pa.array(key_ids_cpu.numpy(), type=pa.int64()),
            pa.ListArray.from_arrays(pa.array(offsets, type=pa.int32()), values),
            pa.repeat(pa.scalar(source, pa.string()), num_rows),
        ], schema=self._schema))
        return num_rows

    def _write_table_chunks(self, table_chunks: list, output_path: str) -> None:
        """Sibling temp file, then os.replace: a kill mid-write never leaves a truncated shard."""
        import pyarrow.parquet as pq

        tmp_path = f"{output_path}.rank{self._rank}.tmp"
        try:
            with pq.ParquetWriter(tmp_path, self._schema) as writer:
                for chunk in (table_chunks or [self._schema.empty_table()]):
                    writer.write_table(chunk)
            os.replace(tmp_path, output_path)
        except BaseException:
            if os.path.exists(tmp_path):
                os.remove(tmp_path)
            raise
