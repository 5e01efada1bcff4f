"""EmbeddingBagCollection on the gfx950 kernels: pooled lookup + in-backward fused sparse optimizer.

Host-side mirror of what sits under ``EmbeddingGroupImpl.forward`` in the reference
(/root/reference/tzrec/modules/embedding.py:855 builds ``EmbeddingBagCollection(list[EmbeddingBag
Config], device)``, :930 calls ``self.ebc(kjt)`` -> ``KeyedTensor``, :972-976 regroups it per feature
group).  Names and argument meanings follow torchrec 1.7.0 (EmbeddingBagConfig.name /
embedding_dim / num_embeddings / feature_names / pooling; ``forward(KeyedJaggedTensor) ->
KeyedTensor`` with keys in table-then-feature order).  Differences by design:

* the optimizer is part of the module (the reference fuses it with
  ``apply_optimizer_in_backward``, /root/reference/tzrec/main.py:774-781): ``loss.backward()``
  updates the rows it touched and leaves no ``.grad`` on the tables;
* ``forward_grouped`` writes the pooled blocks straight into feature-group layout, so the
  regroup copy of embedding.py:972-976 disappears;
* for Adagrad at D=16 a table row is stored as one 128-byte line ``[w(16) | m(16)]``
  (``row_layout="interleaved"``): the backward read-modify-write of a row touches one HBM line.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from . import _lib
from .sparse import KeyedJaggedTensor, KeyedTensor


@dataclass
class EmbeddingBagConfig:
    """torchrec.modules.embedding_configs.EmbeddingBagConfig fields tzrec fills
    (/root/reference/tzrec/features/feature.py:611-636)."""

    name: str
    embedding_dim: int
    num_embeddings: int
    feature_names: List[str] = field(default_factory=list)
    pooling: str = "sum"  # PoolingType.SUM / MEAN
    init_fn: Optional[Callable[[torch.Tensor], None]] = None
    # tzrec sets `.trainable` from the feature config (feature.py:629); frozen tables are left out
    # of the fused optimizer (BaseModel.sparse_parameters, tzrec/models/model.py:162-201)
    trainable: bool = True
    # `data_type` of the feature config (tzrec/features/feature.py:346-356,626): "FP32" | "FP16".
    # FP16 rows are widened to fp32 on read; the fused optimizer computes in fp32 and stores back
    # with round-to-nearest-even; optimizer state stays fp32.
    data_type: str = "FP32"


@dataclass
class SparseOptimizerConfig:
    """Fused sparse optimizer (/root/reference/tzrec/optim/optimizer_builder.py:30-97,
    protos/optimizer.proto:76-139): ``adagrad_optimizer`` -> kind "adagrad",
    ``rowwise_adagrad_optimizer`` -> "rowwise_adagrad", ``sgd_optimizer`` -> "sgd",
    ``adam_optimizer`` -> "adam" (state [rows, 2 D]: exp_avg | exp_avg_sq; one step counter per
    optimizer, bias-corrected as fbgemm's split Adam)."""

    kind: str = "adagrad"
    lr: float = 0.002
    eps: float = 1e-8  # fbgemm default [upstream]; tzrec protos do not expose it
    weight_decay: float = 0.0
    weight_decay_mode: str = "none"  # NONE | L2 | DECOUPLE (rowwise adagrad)
    gradient_clipping: bool = False
    max_gradient: float = 1.0
    initial_accumulator_value: float = 0.0
    beta1: float = 0.9  # adam
    beta2: float = 0.999


_OPT_KIND = {"sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "rowwise_adagrad": _lib.OPT_ROWWISE_ADAGRAD,
             "adam": _lib.OPT_ADAM}
_WD_MODE = {"none": _lib.WD_NONE, "l2": _lib.WD_L2, "decouple": _lib.WD_DECOUPLE}


class FusedSparseOptimizer:
    """KeyedOptimizer-like handle of the in-backward optimizer (``model.fused_optimizer`` in the
    reference, /root/reference/tzrec/main.py:849,877-879): schedulers mutate
    ``param_groups[i]["lr"]``; the value is mirrored into a device scalar the kernels read, so a
    captured hipGraph sees new learning rates without re-capture."""

    def __init__(self, cfg: SparseOptimizerConfig, ebc: "EmbeddingBagCollection") -> None:
        self.cfg = cfg
        self._ebc = ebc
        self.param_groups = [{"lr": float(cfg.lr), "params": list(ebc.table_weights().values())}]
        self._lr_dev: Optional[torch.Tensor] = None
        self._lr_host: Optional[float] = None
        self._adam: Optional[torch.Tensor] = None  # device {step, 1 - b1^step, 1 - b2^step, -}

    def adam_state(self, device: torch.device) -> torch.Tensor:
        if self._adam is None or self._adam.device != device:
            self._adam = torch.zeros(4, dtype=torch.float32, device=device)
        return self._adam

    def begin_step(self, device: torch.device) -> None:
        """Once per training step, before the step's update kernels: advances Adam's step counter on
        the device (one tiny launch, graph-capturable); nothing for the other optimizers."""
        if self.cfg.kind == "adam":
            _lib.check(_lib.lib().tzr_sparse_adam_tick(_lib.ptr(self.adam_state(device)), self.cfg.beta1, self.cfg.beta2,
                                                       _lib.stream_ptr(device)), "tzr_sparse_adam_tick")

    def optim_struct(self, device: torch.device, kind: Optional[int] = None) -> "_lib.TzrSparseOptim":
        """The TzrSparseOptim every update launch of this optimizer passes (`kind` overrides the
        configured one: TZR_OPT_ACCUMULATE for replicated tables)."""
        cfg = self.cfg
        opt = _lib.TzrSparseOptim()
        opt.kind = _OPT_KIND[cfg.kind] if kind is None else kind
        opt.weight_decay_mode = _WD_MODE[cfg.weight_decay_mode.lower()]
        opt.d_lr = _lib.ptr(self.lr_device(device))
        opt.eps, opt.weight_decay, opt.max_gradient = cfg.eps, cfg.weight_decay, cfg.max_gradient
        opt.gradient_clipping = 1 if cfg.gradient_clipping else 0
        opt.beta1, opt.beta2 = cfg.beta1, cfg.beta2
        opt.d_adam = _lib.ptr(self.adam_state(device)) if cfg.kind == "adam" else 0
        return opt

    @property
    def params(self) -> Dict[str, torch.Tensor]:
        return self._ebc.table_weights()

    def lr_device(self, device: torch.device) -> torch.Tensor:
        """The device scalar the update kernels read.  Outside a graph capture it is refreshed from
        ``param_groups[0]["lr"]`` here; UNDER capture nothing is written (a captured ``fill_`` would reset the
        learning rate on every replay): whoever replays graphs calls ``sync_lr`` before each replay."""
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if self._lr_dev is None or self._lr_dev.device != device:
            if capturing:  # (the allocation + fill would be captured and replayed)
                raise RuntimeError("the fused sparse optimizer's device learning rate does not exist yet: run one eager step "
                                   "(or call fused_optimizer.sync_lr(device)) before capturing a graph with update kernels")
            lr = float(self.param_groups[0]["lr"])
            self._lr_dev = torch.full((1,), lr, dtype=torch.float32, device=device)
            self._lr_host = lr
        elif not capturing:
            self.sync_lr(device)
        elif float(self.param_groups[0]["lr"]) != self._lr_host:
            # a scheduler moved the rate and nobody mirrored it: the captured kernels would train with the stale one
            raise RuntimeError("learning rate changed since the last sync and a hipGraph capture is open: call "
                               "dense.sync_learning_rates(model, optimizer) before capturing / replaying (INTEGRATION.md)")
        return self._lr_dev

    def sync_lr(self, device: Optional[torch.device] = None) -> None:
        """Mirror ``param_groups[0]["lr"]`` (which schedulers mutate, /root/reference/tzrec/main.py:877-879) into the
        device scalar.  Call outside capture, before every replay of a graph that holds update kernels."""
        if self._lr_dev is None:
            if device is None:
                return
            self.lr_device(device)
            return
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self._lr_dev.fill_(lr)
            self._lr_host = lr

    def zero_grad(self, set_to_none: bool = True) -> None:  # tables never hold .grad
        pass

    def step(self, closure=None) -> None:  # the update already happened inside backward
        pass

    def state_dict(self) -> Dict[str, object]:
        return {
            "state": {n: {"momentum1": s} for n, s in self._ebc.table_states().items()},
            "param_groups": [{"lr": self.param_groups[0]["lr"]}],
            "adam_step": None if self._adam is None else float(self._adam[0]),
        }

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        for n, st in sd.get("state", {}).items():
            dst = self._ebc.table_states().get(n)
            if dst is not None:
                dst.copy_(st["momentum1"])
        if sd.get("param_groups"):
            self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]
        if sd.get("adam_step") is not None and self.cfg.kind == "adam":
            self.set_adam_step(float(sd["adam_step"]))

    def set_adam_step(self, t: float, device: Optional[torch.device] = None) -> None:
        st = self.adam_state(device or self._ebc._device)
        st.copy_(torch.tensor([t, 1.0 - self.cfg.beta1 ** t, 1.0 - self.cfg.beta2 ** t, 0.0], dtype=torch.float32))


class _Table(nn.Module):
    def __init__(self, weight: torch.Tensor) -> None:
        super().__init__()
        self.weight = nn.Parameter(weight, requires_grad=False)


@dataclass
class _Lookup:
    key: str  # KJT key
    table: int
    out_key: str  # name of the pooled block (feature, or feature@table when ambiguous)


class _Meta:
    """Device-side descriptors for one KJT key list (see include/tzrec_hip.h)."""

    def __init__(self) -> None:
        self.tables_np = self.feats_np = self.slots_np = None
        self.d_tables = self.d_feats = self.d_slots = None
        self.d_bwd_tables = self.d_bwd_feats = None
        self.bwd_tables_np = self.bwd_feats_np = None  # host copies of the backward descriptors (cells geometry)
        self.n_keys = 0
        self.cells: Dict[int, Optional["_CellsGeo"]] = {}  # batch size -> geometry of the one-launch plan (None: not a case for it)


class _CellsGeo:
    """The geometry buffer of the cells plan for one (descriptor set, batch size): host image + device copy (the kernels write
    its tail), and the host's view of its overflow word (include/tzrec_hip.h: tzr_bwd_cells_geometry)."""

    def __init__(self, image: np.ndarray, info, device: torch.device) -> None:
        self.h_img = image                       # uint8, kept: the launchers read the header from it
        self.h_ptr = image.ctypes.data
        self.n_chunks, self.n_units = int(info[1]), int(info[2])
        self.d_img = _lib.workspace(image.nbytes, device)[:image.nbytes]
        self.d_img.copy_(torch.from_numpy(image))
        off = int(info[6])
        self.d_overflow = self.d_img[off:off + 4].view(torch.int32)
        pin = device.type == "cuda"
        self.h_overflow = torch.zeros(1, dtype=torch.int32, pin_memory=pin)
        self.seen = 0          # overflow count already acted on
        self.copied = None     # event behind the last queued copy of the word
        self.launches = 0
        self.demoted = False


def mask_frozen_descriptors(tables: np.ndarray, feats: np.ndarray, frozen: Sequence[bool]):
    """Backward twins of (TzrTable[], TzrFeature[]) in which the lookups of frozen tables
    (`trainable: false`, tzrec/features/feature.py:629, models/model.py:162-201) are marked "not owned"
    (table -1, ordered last): the plan sorts nothing for them and the fused optimizer never touches
    their rows.  The live lookups keep their relative (table-major) order."""
    bt, bf = tables.copy(), feats.copy()
    by_order = np.argsort(feats["order"], kind="stable")
    live = [int(i) for i in by_order if feats[i]["table"] >= 0 and not frozen[int(feats[i]["table"])]]
    dead = [int(i) for i in by_order if not (feats[i]["table"] >= 0 and not frozen[int(feats[i]["table"])])]
    for o, i in enumerate(live + dead):
        bf[i]["order"] = o
    for i in dead:
        bf[i]["table"] = -1
    for t in range(len(bt)):
        mine = [i for i in live if int(feats[i]["table"]) == t]
        bt[t]["first_order"] = int(bf[mine[0]]["order"]) if mine else 0
        bt[t]["n_feats"] = len(mine)
    return bt, bf


class _PooledLookupFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ebc, kjt, dst_names, hook):  # hook: zero-size tensor that requires grad
        outs = ebc._launch_forward(kjt, dst_names, with_plan=True)
        ctx.ebc, ctx.kjt, ctx.dst_names = ebc, kjt, dst_names
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.ebc._launch_backward(ctx.kjt, ctx.dst_names, grads)
        return None, None, None, None


class EmbeddingBagCollection(nn.Module):
    """Pooled embedding lookup for a set of tables.

    Args:
        tables: table configs; ``feature_names`` are the KJT keys each table serves.
        device: where the table storage lives (a HIP device in production).
        optimizer: fused sparse optimizer; ``None`` makes the tables frozen (forward only).
        groups: optional ``{group_name: [out_key, ...]}`` feature-group layout for
            ``forward_grouped`` (out_key = feature name, or ``feature@table`` when the feature is
            served by several tables).
        row_layout: "interleaved" stores Adagrad state next to the weights in one row
            (``[w | m]``), "split" keeps two arrays.
    """

    def __init__(
        self,
        tables: Sequence[EmbeddingBagConfig],
        device: Optional[torch.device] = None,
        optimizer: Optional[SparseOptimizerConfig] = None,
        groups: Optional[Dict[str, List[str]]] = None,
        row_layout: str = "interleaved",
    ) -> None:
        super().__init__()
        device = torch.device(device) if device is not None else torch.device("cpu")
        self._device = device
        self._configs = list(tables)
        self._opt_cfg = optimizer
        self._row_layout = row_layout
        self.embedding_bags = nn.ModuleDict()
        self._storage: List[torch.Tensor] = []
        self._states: Dict[str, torch.Tensor] = {}
        self._lookups: List[_Lookup] = []
        names = set()
        feat_tables: Dict[str, List[int]] = {}
        for t, cfg in enumerate(self._configs):
            if cfg.name in names:
                raise ValueError(f"duplicate table name {cfg.name}")
            names.add(cfg.name)
            if cfg.embedding_dim % 4 or cfg.embedding_dim > 256:
                raise ValueError(f"{cfg.name}: embedding_dim must be a multiple of 4, <= 256")
            for f in cfg.feature_names:
                feat_tables.setdefault(f, []).append(t)
        # torchrec order: table-then-feature; a feature on >1 tables is exposed as feature@table
        # (/root/reference/tzrec/modules/embedding.py:753-758,826-827)
        for t, cfg in enumerate(self._configs):
            for f in cfg.feature_names:
                out_key = f if len(feat_tables[f]) == 1 else f"{f}@{cfg.name}"
                self._lookups.append(_Lookup(f, t, out_key))
        self._out_dim = {lk.out_key: self._configs[lk.table].embedding_dim for lk in self._lookups}
        self._allocate()
        self._has_fp16 = any(m.weight.dtype == torch.float16 for m in self.embedding_bags.values())
        self._groups = groups
        self._dst_layouts: Dict[Tuple[str, ...], List[Tuple[str, List[str]]]] = {}
        self._meta_cache: Dict[Tuple, _Meta] = {}
        self.fused_optimizer = FusedSparseOptimizer(optimizer, self) if optimizer is not None else None
        self._hook = torch.zeros(0, requires_grad=True, device=device)
        self._timers = None  # bench.py: object with .start(name) -> event recorded after the launch
        self._side_stream = None
        self._lookup_trackers: list = []  # register_post_lookup_tracker_fn
        self._track_segs: Dict[Tuple[str, ...], tuple] = {}
        # The backward index plan only needs the ids, so it could overlap the forward on a side stream
        # (plan_backward_async).  Off by default: with the round-2 plan kernels the side-stream plan came
        # out wrong (not a permutation of the lookups) in ~1/3 of the iterations of scripts/zipf_debug.py
        # whenever other kernels of the main stream ran next to it, while the same kernels in ONE stream
        # were right in 220 of 220 (NOTES.md "side-stream plan"); and the plan is ~60 us of a step now.
        self.async_plan = False
        # ... what does pay: the plan's workgroups INSIDE the forward's launch (tzr_pooled_fwd_cells_plan; batches of one id per bag
        # that take the cells plan).  TZR_FWD_PLAN=0: the two launches.
        self.forward_plan = os.environ.get("TZR_FWD_PLAN", "1") != "0"
        self.forward_plans = 0  # launches that carried a plan
        # a row with more lookups than a workgroup's LDS holds is expected (the shared row of a zero-collision hash's unseen ids, a
        # default id): the one-launch backward then lets all of a table's workgroups sum such a row (TZR_GRAD_HOT_ROWS,
        # include/tzrec_hip.h); zch.ManagedCollisionEmbeddingBagCollection sets it
        self.expect_hot_rows = False
        # Which index plan a batch of one id per bag takes (batches with jagged bags always take the exact one):
        #   "exact"  tzr_pooled_bwd_plan: four launches, any id distribution at full speed (heavy buckets, hot rows);
        #   "cells"  tzr_pooled_bwd_cells_plan: ONE launch; unit sizes are expectations for evenly drawn ids, a unit that holds
        #            more is still right but slow, and counted;
        #   "auto"   cells until that count moves -- the library's overflow word, copied to the host behind every apply and
        #            looked at before the next plan (the first CELLS_PROBATION batches wait for it: a skewed id distribution
        #            is found out before anything is captured into a hipGraph) -- then exact for good.
        self.plan_mode = os.environ.get("TZR_BWD_PLAN", "auto")
        if self.plan_mode not in ("auto", "cells", "exact"):
            raise ValueError("TZR_BWD_PLAN: auto | cells | exact")

    CELLS_PROBATION = 4

    def _cells_geometry(self, meta: _Meta, B: int) -> Optional[_CellsGeo]:
        if B in meta.cells:
            return meta.cells[B]
        L = _lib.lib()
        _, max_dim = self._bwd_dims()
        info = (C.c_int64 * 8)()
        tp, fp = meta.bwd_tables_np.ctypes.data, meta.bwd_feats_np.ctypes.data
        rc = L.tzr_bwd_cells_geometry(tp, len(self._configs), fp, len(self._lookups), B, max_dim, None, 0, info)
        geo = None
        if rc == _lib.TZR_OK:
            img = np.zeros(int(info[0]), dtype=np.uint8)
            _lib.check(L.tzr_bwd_cells_geometry(tp, len(self._configs), fp, len(self._lookups), B, max_dim, img.ctypes.data, img.nbytes,
                                                info), "tzr_bwd_cells_geometry")
            geo = _CellsGeo(img, info, self._device)
        elif rc != -4:  # (TZR_ERR_UNSUPPORTED: simply not a case for the cells plan)
            _lib.check(rc, "tzr_bwd_cells_geometry")
        meta.cells[B] = geo
        return geo

    def reset_plan_mode(self) -> None:
        """Forget what "auto" has learnt about the id distribution (a collection that met skewed ids stays on the exact plan for
        good): for callers that know the distribution has changed -- bench.py times Zipf ids and evenly drawn ones on one
        collection."""
        for meta in self._meta_cache.values():
            for geo in meta.cells.values():
                if geo is not None:
                    if geo.copied is not None:
                        geo.copied.synchronize()
                    geo.seen, geo.demoted, geo.launches = int(geo.h_overflow[0]), False, 0

    def _cells_for(self, kjt: KeyedJaggedTensor, meta: _Meta) -> Optional[_CellsGeo]:
        """the geometry of the one-launch plan when this batch takes it, else None"""
        if self.plan_mode == "exact" or kjt.uniform_length() != 1 or kjt.weights_or_none() is not None:
            return None
        geo = self._cells_geometry(meta, kjt.stride())
        if geo is None or self.plan_mode == "cells":
            return geo
        if geo.demoted:
            return None
        capturing = self._device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if geo.copied is not None and geo.launches <= self.CELLS_PROBATION and not capturing:
            geo.copied.synchronize()  # (the first batches: know before deciding; later the word is read as it lies)
        if int(geo.h_overflow[0]) != geo.seen:
            geo.demoted = True  # units overflowed: ids this skewed belong on the exact plan (heavy buckets, hot rows)
            return None
        return geo

    # -- storage ---------------------------------------------------------------------------
    def _allocate(self) -> None:
        kind = self._opt_cfg.kind if self._opt_cfg is not None else None
        init_m = self._opt_cfg.initial_accumulator_value if self._opt_cfg is not None else 0.0
        for cfg in self._configs:
            rows, D = cfg.num_embeddings, cfg.embedding_dim
            dt = cfg.data_type.upper()
            if dt not in ("FP32", "FP16"):
                raise ValueError(f"{cfg.name}: data_type {cfg.data_type!r} not supported (FP32 | FP16)")
            if dt == "FP16":  # half weights, fp32 state, never interleaved (elements differ in size)
                store = torch.empty(rows, D, dtype=torch.float16, device=self._device)
                weight = store
                state = (torch.full((rows, D), init_m, dtype=torch.float32, device=self._device) if kind == "adagrad"
                         else torch.zeros(rows, dtype=torch.float32, device=self._device) if kind == "rowwise_adagrad"
                         else torch.zeros(rows, 2 * D, dtype=torch.float32, device=self._device) if kind == "adam" else None)
            elif kind == "adagrad" and self._row_layout == "interleaved":
                store = torch.empty(rows, 2 * D, dtype=torch.float32, device=self._device)
                weight, state = store[:, :D], store[:, D:]
                state.fill_(init_m)
            elif kind == "rowwise_adagrad" and self._row_layout == "interleaved":
                # [w(D) | m | pad]: the row's scalar state sits in the 64-byte sector NEXT to its weights (same
                # 128-byte line and DRAM page at D = 16) instead of in a [rows] array of its own, where every touched
                # row cost a second, unrelated sector fetch + a 4-byte write into a third line.  Twice the table
                # bytes (as the interleaved Adagrad layout): HBM is sized for it (26 GB for DLRM-Criteo of 288 GB).
                store = torch.zeros(rows, 2 * D, dtype=torch.float32, device=self._device)
                weight, state = store[:, :D], store[:, D]
            else:
                store = torch.empty(rows, D, dtype=torch.float32, device=self._device)
                weight = store
                if kind == "adagrad":
                    state = torch.full((rows, D), init_m, dtype=torch.float32, device=self._device)
                elif kind == "rowwise_adagrad":
                    state = torch.zeros(rows, dtype=torch.float32, device=self._device)
                elif kind == "adam":  # [exp_avg | exp_avg_sq]
                    state = torch.zeros(rows, 2 * D, dtype=torch.float32, device=self._device)
                else:
                    state = None
            if cfg.init_fn is not None:
                cfg.init_fn(weight)
            else:  # EmbeddingBagConfig default [upstream]: uniform(-sqrt(1/rows), sqrt(1/rows))
                a = math.sqrt(1.0 / max(rows, 1))
                weight.uniform_(-a, a)
            self._storage.append(store)
            self.embedding_bags[cfg.name] = _Table(weight)
            if state is not None:
                self._states[cfg.name] = state

    def table_weights(self) -> Dict[str, torch.Tensor]:
        return {n: m.weight for n, m in self.embedding_bags.items()}

    def table_states(self) -> Dict[str, torch.Tensor]:
        return self._states

    def embedding_bag_configs(self) -> List[EmbeddingBagConfig]:
        return self._configs

    @property
    def device(self) -> torch.device:
        return self._device

    # -- descriptors -------------------------------------------------------------------------
    def _default_layout(self) -> List[Tuple[str, List[str]]]:
        return [("__all__", [lk.out_key for lk in self._lookups])]

    def _meta(self, kjt_keys: Sequence[str], layout: List[Tuple[str, List[str]]]) -> _Meta:
        ck = (tuple(kjt_keys), tuple((n, tuple(ks)) for n, ks in layout),
              tuple(m.weight.data_ptr() for m in self.embedding_bags.values()))
        meta = self._meta_cache.get(ck)
        if meta is not None:
            return meta
        if len(layout) > _lib.TZR_MAX_DST:
            raise ValueError(f"at most {_lib.TZR_MAX_DST} feature groups per lookup")
        key_index = {k: i for i, k in enumerate(kjt_keys)}
        T, Fn = len(self._configs), len(self._lookups)
        feats = np.zeros(Fn, dtype=_lib.FEATURE_DT)
        feats["dst"] = -1
        by_out = {lk.out_key: i for i, lk in enumerate(self._lookups)}
        for i, lk in enumerate(self._lookups):
            if lk.key not in key_index:
                raise KeyError(f"KeyedJaggedTensor has no key {lk.key!r} needed by table "
                               f"{self._configs[lk.table].name!r}")
            feats[i]["table"] = lk.table
            feats[i]["key"] = key_index[lk.key]
            feats[i]["pooling"] = _lib.POOL_MEAN if self._configs[lk.table].pooling.lower() == "mean" else _lib.POOL_SUM
        # lookups are already in (table, index) order
        feats["order"] = np.arange(Fn, dtype=np.int32)
        slots = []
        for d, (_, out_keys) in enumerate(layout):
            col = 0
            for ok in out_keys:
                i = by_out[ok]
                n = int(feats[i]["n_dst"])
                if n >= _lib.TZR_MAX_FEAT_DST:
                    raise ValueError(f"{ok}: a lookup can be copied into at most {_lib.TZR_MAX_FEAT_DST} groups")
                feats[i]["dst"][n] = d
                feats[i]["col"][n] = col
                feats[i]["n_dst"] = n + 1
                D = self._configs[self._lookups[i].table].embedding_dim
                for c in range(D // 4):
                    slots.append((i, c, d, col + 4 * c))
                col += D
        tables = np.zeros(T, dtype=_lib.TABLE_DT)
        kind = self._opt_cfg.kind if self._opt_cfg is not None else None
        for t, cfg in enumerate(self._configs):
            w = self.embedding_bags[cfg.name].weight
            st = self._states.get(cfg.name)
            tables[t]["w"] = w.data_ptr()
            tables[t]["m"] = st.data_ptr() if st is not None else 0
            tables[t]["rows"] = cfg.num_embeddings
            tables[t]["dim"] = cfg.embedding_dim
            tables[t]["w_stride"] = w.stride(0)
            tables[t]["w_dtype"] = _lib.DT_F16 if w.dtype == torch.float16 else _lib.DT_F32
            tables[t]["m_stride"] = st.stride(0) if st is not None else 0  # row-wise Adagrad: one float per row
            mine = [i for i, lk in enumerate(self._lookups) if lk.table == t]
            tables[t]["first_order"] = mine[0] if mine else 0
            tables[t]["n_feats"] = len(mine)
        meta = _Meta()
        meta.tables_np, meta.feats_np = tables, feats
        meta.slots_np = np.array(slots, dtype=_lib.SLOT_DT)
        meta.d_tables = _lib.upload_struct(tables, self._device)
        meta.d_feats = _lib.upload_struct(feats, self._device)
        # backward descriptors: lookups of frozen tables are marked "not owned" (table -1, ordered
        # last), so the plan sorts nothing for them and the optimizer never touches their rows
        frozen = [not cfg.trainable for cfg in self._configs]
        meta.d_bwd_tables, meta.d_bwd_feats = meta.d_tables, meta.d_feats
        meta.bwd_tables_np, meta.bwd_feats_np = tables, feats
        if any(frozen):
            bt, bf = mask_frozen_descriptors(tables, feats, frozen)
            meta.bwd_tables_np, meta.bwd_feats_np = bt, bf
            meta.d_bwd_tables = _lib.upload_struct(bt, self._device)
            meta.d_bwd_feats = _lib.upload_struct(bf, self._device)
        meta.d_slots = _lib.upload_struct(meta.slots_np, self._device)
        meta.n_keys = len(kjt_keys)
        self._meta_cache[ck] = meta
        return meta

    def _layout_for(self, dst_names: Tuple[str, ...]) -> List[Tuple[str, List[str]]]:
        if dst_names == ("__all__",):
            return self._default_layout()
        assert self._groups is not None
        return [(g, self._groups[g]) for g in dst_names]

    # -- launches ----------------------------------------------------------------------------
    def _kjt_args(self, kjt: KeyedJaggedTensor):
        uniform = kjt.uniform_length() == 1
        offsets = None if uniform else kjt.offsets()
        return uniform, offsets

    def _forward_carries_plan(self, kjt: KeyedJaggedTensor, meta: _Meta) -> Optional["_CellsGeo"]:
        """The cells geometry when this batch's forward can carry the backward's index plan in its launch
        (tzr_pooled_fwd_cells_plan: one id per bag, fp32 tables, no per-sample weights, a batch the cells plan takes)."""
        if not self.forward_plan or self._has_fp16 or kjt.uniform_length() != 1 or kjt.weights_or_none() is not None:
            return None
        cached = getattr(kjt, "_tzr_plan", None)
        if cached is not None and not (len(cached) > 5 and cached[5] == "forward"):
            return None  # planned ahead by the caller (plan_backward / plan_backward_async): that plan is the batch's
        # (a plan left by an earlier forward of this object whose backward never ran is dropped: the ids may have been refreshed in place)
        if self.backward_is_direct(kjt):
            return None
        if not _lib.lib().tzr_pooled_fwd_cells_plan_supported(len(meta.slots_np), kjt.stride()):
            return None
        return self._cells_for(kjt, meta)

    def _launch_forward(self, kjt: KeyedJaggedTensor, dst_names: Tuple[str, ...], with_plan: bool = False) -> List[torch.Tensor]:
        """`with_plan`: a backward of this batch follows (the autograd node's forward): the launch may carry its index plan."""
        layout = self._layout_for(dst_names)
        meta = self._meta(kjt.keys(), layout)
        B = kjt.stride()
        uniform, offsets = self._kjt_args(kjt)
        widths = [sum(self._out_dim[k] for k in ks) for _, ks in layout]
        outs = [torch.empty(B, w, dtype=torch.float32, device=self._device) for w in widths]
        dsts = (_lib.TzrDst * len(outs))()
        for i, o in enumerate(outs):
            dsts[i].ptr = _lib.ptr(o)
            dsts[i].stride = o.stride(0)
        geo = self._forward_carries_plan(kjt, meta) if with_plan else None
        if geo is not None:
            L = _lib.lib()
            N = kjt.values().numel()
            max_dim = self._bwd_dims()[1]
            ws = _lib.workspace(L.tzr_pooled_bwd_workspace(N, self._n_positions(kjt), len(self._lookups), len(self._configs), B, max_dim),
                                self._device)
            ev = self._timers.start("fwd+plan") if self._timers is not None else None
            rc = L.tzr_pooled_fwd_cells_plan(
                _lib.ptr(meta.d_tables), _lib.ptr(meta.d_feats), len(self._lookups), _lib.ptr(meta.d_slots), len(meta.slots_np), dsts,
                len(outs), _lib.ptr(meta.d_bwd_tables), len(self._configs), _lib.ptr(meta.d_bwd_feats), len(self._lookups), max_dim,
                _lib.ptr(kjt.values()), N, B, geo.h_ptr, _lib.ptr(geo.d_img), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self._device))
            if rc != -4:  # (TZR_ERR_UNSUPPORTED: nothing was launched -- the two calls, as below)
                if ev is not None:
                    ev.record()
                _lib.check(rc, "tzr_pooled_fwd_cells_plan")
                if self._device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                    ws.record_stream(torch.cuda.current_stream(self._device))
                kjt._tzr_plan = (id(self), dst_names, ws, None, geo, "forward")  # type: ignore[attr-defined]
                self.forward_plans += 1
                return outs
            if self._timers is not None:
                self._timers.pairs["fwd+plan"].pop()
        ev = self._timers.start("fwd") if self._timers is not None else None
        rc = _lib.lib().tzr_pooled_fwd_ex(
            _lib.ptr(meta.d_tables), _lib.ptr(meta.d_feats), len(self._lookups),
            _lib.ptr(meta.d_slots), len(meta.slots_np), _lib.ptr(kjt.values()), _lib.ptr(offsets),
            _lib.ptr(kjt.weights_or_none()), B, dsts, len(outs), 1 if uniform else 0,
            _lib.FWD_MIXED_DTYPE if self._has_fp16 else 0, _lib.stream_ptr(self._device),
        )
        if ev is not None:
            ev.record()
        _lib.check(rc, "tzr_pooled_fwd")
        return outs

    def _bwd_dims(self) -> Tuple[int, int]:
        return (max(c.num_embeddings for c in self._configs), max(c.embedding_dim for c in self._configs))

    def _n_positions(self, kjt: KeyedJaggedTensor) -> int:
        """Capacity of the backward plan's table-major position space: every lookup (key -> table)
        contributes its key's ids, so a key read through k tables counts k times."""
        if kjt.uniform_length() is not None:
            return len(self._lookups) * kjt.stride() * kjt.uniform_length()
        mult: Dict[str, int] = {}
        for lk in self._lookups:
            mult[lk.key] = mult.get(lk.key, 0) + 1
        return kjt.values().numel() * max(mult.values())

    def _direct_workspace(self, n_positions: int) -> torch.Tensor:
        """The persistent workspace of tzr_pooled_bwd_direct (zeroed once: its arrival counters reset themselves; the
        launches of one collection run in stream order, so one buffer per size serves them all)."""
        cache = self.__dict__.setdefault("_direct_ws", {})
        ws = cache.get(n_positions)
        if ws is None:
            _, max_dim = self._bwd_dims()
            # (never evicted: captured hipGraphs replay launches that hold these addresses)
            ws = cache[n_positions] = _lib.zeroed_workspace(
                _lib.lib().tzr_pooled_bwd_direct_workspace(n_positions, len(self._configs), max_dim), self._device)
        return ws

    def backward_is_direct(self, kjt: KeyedJaggedTensor) -> bool:
        """Small batches skip the index plan: ONE launch sorts and applies (tzr_pooled_bwd_direct,
        csrc/pooled_bwd_direct.hip) -- when the library takes the shape and the size (tzr_tune "bwd_direct")."""
        return bool(_lib.lib().tzr_pooled_bwd_direct_supported(self._n_positions(kjt), len(self._lookups), len(self._configs),
                                                               1 if kjt.uniform_length() == 1 else 0, 0))

    def backward_form(self, kjt: KeyedJaggedTensor, dst_names: Tuple[str, ...] = ("__all__",)) -> str:
        """which form the fused backward of this batch takes right now: "direct" (one launch, no plan), "cells" (one-launch plan +
        apply) or "exact" (four-launch plan + apply)"""
        if self.backward_is_direct(kjt):
            return "direct"
        meta = self._meta(kjt.keys(), self._layout_for(dst_names))
        return "cells" if self._cells_for(kjt, meta) is not None else "exact"

    def plan_backward(self, kjt: KeyedJaggedTensor, dst_names: Tuple[str, ...] = ("__all__",)) -> Optional[torch.Tensor]:
        """K6: build the backward index plan for this batch (depends on ids only, so callers may run
        it early on a side stream).  Returns the workspace holding the plan -- None for a batch whose
        backward needs none (`backward_is_direct`)."""
        if self.backward_is_direct(kjt):
            return None
        layout = self._layout_for(dst_names)
        meta = self._meta(kjt.keys(), layout)
        L = _lib.lib()
        B, N = kjt.stride(), kjt.values().numel()
        uniform, offsets = self._kjt_args(kjt)
        max_rows, max_dim = self._bwd_dims()
        NP = self._n_positions(kjt)
        nbytes = L.tzr_pooled_bwd_workspace(N, NP, len(self._lookups), len(self._configs), B, max_dim)
        ws = _lib.workspace(nbytes, self._device)
        geo = self._cells_for(kjt, meta)
        ev = self._timers.start("plan") if self._timers is not None else None
        if geo is not None:
            rc = L.tzr_pooled_bwd_cells_plan(
                _lib.ptr(meta.d_bwd_tables), len(self._configs), _lib.ptr(meta.d_bwd_feats), len(self._lookups), max_dim,
                _lib.ptr(kjt.values()), N, B, geo.h_ptr, _lib.ptr(geo.d_img), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self._device))
        else:
            rc = L.tzr_pooled_bwd_plan(
                _lib.ptr(meta.d_bwd_tables), len(self._configs), _lib.ptr(meta.d_bwd_feats), len(self._lookups),
                meta.n_keys, max_rows, max_dim, _lib.ptr(kjt.values()), _lib.ptr(offsets), N, NP, B,
                1 if uniform else 0, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self._device),
            )
        if ev is not None:
            ev.record()
        _lib.check(rc, "tzr_pooled_bwd_cells_plan" if geo is not None else "tzr_pooled_bwd_plan")
        if self._device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            ws.record_stream(torch.cuda.current_stream(self._device))
        kjt._tzr_plan = (id(self), dst_names, ws, None, geo)  # type: ignore[attr-defined]
        return ws

    def plan_backward_async(self, kjt: KeyedJaggedTensor, dst_names: Tuple[str, ...] = ("__all__",)) -> None:
        """Run K6 on a side HIP stream so it overlaps the forward and the dense MLPs (the reference's
        TrainPipelineSparseDist runs the input dist of the next batch on its own stream for the same
        reason, /root/reference/tzrec/utils/dist_util.py:221-303).  The backward waits on the event."""
        if self._device.type != "cuda" or self.backward_is_direct(kjt):
            self.plan_backward(kjt, dst_names)
            return
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self._device)
        cur = torch.cuda.current_stream(self._device)
        self._side_stream.wait_stream(cur)  # ids / descriptors were produced on the current stream
        with torch.cuda.stream(self._side_stream):
            ws = self.plan_backward(kjt, dst_names)
            ev = torch.cuda.Event()
            ev.record()
        if not torch.cuda.is_current_stream_capturing():  # (a captured step owns its buffers for the graph's life)
            for t in (kjt.values(), kjt.offsets_or_none()):
                if t is not None:
                    t.record_stream(self._side_stream)
        kjt._tzr_plan = (id(self), dst_names, ws, ev, kjt._tzr_plan[4])  # type: ignore[attr-defined]

    def _launch_backward(self, kjt: KeyedJaggedTensor, dst_names: Tuple[str, ...], grads) -> None:
        if self.fused_optimizer is None:
            return  # frozen tables
        direct = self.backward_is_direct(kjt)
        cached = getattr(kjt, "_tzr_plan", None)
        geo = None
        if direct:
            ws = None
        elif cached is not None and cached[0] == id(self) and cached[1] == dst_names:
            ws, geo = cached[2], cached[4]
            if cached[3] is not None:
                torch.cuda.current_stream(self._device).wait_event(cached[3])
                if not torch.cuda.is_current_stream_capturing():
                    ws.record_stream(torch.cuda.current_stream(self._device))
        else:
            ws = self.plan_backward(kjt, dst_names)
            geo = kjt._tzr_plan[4]  # type: ignore[attr-defined]
        layout = self._layout_for(dst_names)
        meta = self._meta(kjt.keys(), layout)
        B, N = kjt.stride(), kjt.values().numel()
        uniform, offsets = self._kjt_args(kjt)
        max_rows, max_dim = self._bwd_dims()
        widths = [sum(self._out_dim[k] for k in ks) for _, ks in layout]
        gl = []
        for g, w in zip(grads, widths):
            if g is None:
                g = torch.zeros(B, w, dtype=torch.float32, device=self._device)
            g = g.contiguous()
            if g.dtype != torch.float32:
                g = g.float()
            gl.append(g)
        gd = (_lib.TzrDst * len(gl))()
        for i, g in enumerate(gl):
            gd[i].ptr = _lib.ptr(g)
            gd[i].stride = g.stride(0)
        self.fused_optimizer.begin_step(self._device)
        opt = self.fused_optimizer.optim_struct(self._device)
        ev = self._timers.start("apply") if self._timers is not None else None
        if direct:
            dws = self._direct_workspace(self._n_positions(kjt))
            rc = _lib.lib().tzr_pooled_bwd_direct(
                _lib.ptr(meta.d_bwd_tables), len(self._configs), _lib.ptr(meta.d_bwd_feats), len(self._lookups), max_rows, max_dim,
                _lib.ptr(kjt.values()), _lib.ptr(offsets), _lib.ptr(kjt.weights_or_none()), N, self._n_positions(kjt), B,
                1 if uniform else 0, _lib.GRAD_HOT_ROWS if self.expect_hot_rows else 0, gd, len(gl), opt, _lib.ptr(dws), dws.numel(),
                _lib.stream_ptr(self._device))
        elif geo is not None:
            rc = _lib.lib().tzr_pooled_bwd_cells_apply(
                _lib.ptr(meta.d_bwd_tables), _lib.ptr(meta.d_bwd_feats), len(self._lookups), len(self._configs), max_dim, None, N, B, 0,
                gd, len(gl), opt, geo.h_ptr, _lib.ptr(geo.d_img), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self._device))
        else:
            rc = _lib.lib().tzr_pooled_bwd_apply(
                _lib.ptr(meta.d_bwd_tables), _lib.ptr(meta.d_bwd_feats), len(self._lookups), len(self._configs),
                max_dim, _lib.ptr(offsets), _lib.ptr(kjt.weights_or_none()), N, self._n_positions(kjt), B,
                1 if uniform else 0, 0,
                gd, len(gl), opt, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(self._device),
            )
        if ev is not None:
            ev.record()
        _lib.check(rc, "tzr_pooled_bwd_direct" if direct else ("tzr_pooled_bwd_cells_apply" if geo is not None else "tzr_pooled_bwd_apply"))
        if geo is not None and self.plan_mode == "auto":
            # the overflow word follows the apply to the host (8 bytes, no wait): `_cells_for` looks at it before the next plan
            geo.launches += 1
            geo.h_overflow.copy_(geo.d_overflow, non_blocking=True)
            if self._device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
                geo.copied = torch.cuda.Event()
                geo.copied.record(torch.cuda.current_stream(self._device))
        kjt._tzr_plan = None  # type: ignore[attr-defined]

    def register_post_lookup_tracker_fn(self, fn) -> None:
        """torchrec's hook of the same name, which the reference's ModelDeltaTracker registers on every
        sharded collection (/root/reference/tzrec/utils/delta_embedding_dump.py:423-427).  After each
        lookup `fn(self, segs, ids, key_offsets, key_stride, uniform_len)` is called on the lookup's
        stream: lookup segment i reads table `segs[i][0]` with the ids of key segment `segs[i][1]`
        (addressing as in `tzr_delta_mark`)."""
        self._lookup_trackers.append(fn)

    def _notify_trackers(self, kjt: KeyedJaggedTensor) -> None:
        keys = tuple(kjt.keys())
        segs = self._track_segs.get(keys)
        if segs is None:
            index = {k: i for i, k in enumerate(keys)}
            segs = tuple((self._configs[lk.table].name, index[lk.key]) for lk in self._lookups if lk.key in index)
            self._track_segs[keys] = segs
        uniform, offsets = self._kjt_args(kjt)
        for fn in self._lookup_trackers:
            fn(self, segs, kjt.values(), offsets, kjt.stride(), 1 if uniform else 0)

    def _run(self, kjt: KeyedJaggedTensor, dst_names: Tuple[str, ...]) -> List[torch.Tensor]:
        # a batch whose KeyedJaggedTensor holds sequence keys behind this collection's one-id-per-bag keys: the uniform view of
        # the keys in front (sparse.uniform_prefix) -- the forward's one-id form, the one-launch / cells backward
        if kjt.uniform_length() is None and getattr(kjt, "_uniform_keys", None):
            view = kjt.uniform_prefix([lk.key for lk in self._lookups])
            if view is not None:
                kjt = view
        if self._lookup_trackers:
            self._notify_trackers(kjt)
        if torch.is_grad_enabled() and self.fused_optimizer is not None and self.training:
            if self.async_plan and getattr(kjt, "_tzr_plan", None) is None:
                self.plan_backward_async(kjt, dst_names)
            return list(_PooledLookupFn.apply(self, kjt, dst_names, self._hook))
        return self._launch_forward(kjt, dst_names)

    # -- public API --------------------------------------------------------------------------
    def forward(self, features: KeyedJaggedTensor) -> KeyedTensor:
        """``self.ebc(kjt)`` of the reference: KeyedTensor [B, sum D] in table-then-feature order."""
        (out,) = self._run(features, ("__all__",))
        keys = [lk.out_key for lk in self._lookups]
        return KeyedTensor(keys, [self._out_dim[k] for k in keys], out)

    def forward_grouped(self, features: KeyedJaggedTensor, group_names: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
        """Pooled lookup written directly in feature-group layout (fuses regroup_as_dict)."""
        if self._groups is None:
            raise ValueError("EmbeddingBagCollection was built without groups")
        names = tuple(group_names) if group_names is not None else tuple(self._groups)
        outs = self._run(features, names)
        return dict(zip(names, outs))

    def bounds_check(self, features: KeyedJaggedTensor, mode: int = _lib.BOUNDS_WARNING) -> torch.Tensor:
        """K4; returns the device counter (int64[1]) of out-of-range ids."""
        meta = self._meta(features.keys(), self._default_layout())
        cnt = torch.zeros(1, dtype=torch.int64, device=self._device)
        rc = _lib.lib().tzr_bounds_check(
            _lib.ptr(meta.d_tables), _lib.ptr(meta.d_feats), len(self._lookups),
            _lib.ptr(features.values()), _lib.ptr(features.offsets()), features.stride(), mode,
            _lib.ptr(cnt), _lib.stream_ptr(self._device),
        )
        _lib.check(rc, "tzr_bounds_check")
        return cnt
