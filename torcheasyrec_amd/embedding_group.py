"""EmbeddingGroup: features + feature groups -> tables -> grouped tensors.

Host-side mirror of ``tzrec.modules.embedding.EmbeddingGroup / EmbeddingGroupImpl`` for pooled
(DEEP / WIDE) groups (/root/reference/tzrec/modules/embedding.py:167-536, 681-978):

* table name = ``embedding_name`` or ``{feature}_emb`` (features/feature.py:615); WIDE groups append
  ``_wide`` and force dim = ``wide_embedding_dim or 4`` (embedding.py:744-745,777-786); an optional
  ``_{embedding_name_suffix}`` follows (:746-747);
* tables with one name are shared iff rows, dim and pooling match, else an error (:576-600);
* a feature that lands on more than one table is exposed as ``feature@table`` (:753-758,826-827);
* a group's tensor is the column concat of its features in ``feature_names`` order; dense (raw)
  features are columns of the dense KeyedTensor (:963-976).

``forward(batch)`` returns ``{group_name: Tensor}`` like the reference (rank_model.py:116); the
pooled blocks are written in group layout by the lookup kernel, so no regroup copy happens for
groups made only of sparse features.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from .config import FeatureGroupSpec, FeatureSpec
from .embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
from .sparse import KeyedJaggedTensor, KeyedTensor

BASE_DATA_GROUP = "__BASE__"  # tzrec/datasets/utils.py:28


@dataclass
class Batch:
    """Input data batch (tzrec/datasets/utils.py:298-463): dense KeyedTensor, sparse KJT and labels
    per data group; Pipelineable (to / record_stream / pin_memory)."""

    dense_features: Dict[str, KeyedTensor] = field(default_factory=dict)
    sparse_features: Dict[str, KeyedJaggedTensor] = field(default_factory=dict)
    labels: Dict[str, torch.Tensor] = field(default_factory=dict)
    sample_weights: Dict[str, torch.Tensor] = field(default_factory=dict)
    # multi-valued sequence features (value_dim != 1): per data group a KJT whose values are the ids
    # per sequence STEP and whose lengths are the steps per sample (tzrec/datasets/data_parser.py:554-593)
    sequence_mulval_lengths: Dict[str, KeyedJaggedTensor] = field(default_factory=dict)
    # dense (raw) sub-features of a sequence_feature block: feature name -> sequence.JaggedTensor with
    # values [sum of steps, value_dim] and lengths = steps per sample (data_parser.py:450-456)
    sequence_dense_features: Dict[str, object] = field(default_factory=dict)

    def to(self, device, non_blocking: bool = False) -> "Batch":
        return Batch(
            {k: v.to(device, non_blocking) for k, v in self.dense_features.items()},
            {k: v.to(device, non_blocking) for k, v in self.sparse_features.items()},
            {k: v.to(device, non_blocking=non_blocking) for k, v in self.labels.items()},
            {k: v.to(device, non_blocking=non_blocking) for k, v in self.sample_weights.items()},
            {k: v.to(device, non_blocking) for k, v in self.sequence_mulval_lengths.items()},
            {k: v.to(device, non_blocking) for k, v in self.sequence_dense_features.items()},
        )

    def record_stream(self, stream) -> None:
        for v in self.dense_features.values():
            v.record_stream(stream)
        for v in (list(self.sparse_features.values()) + list(self.sequence_mulval_lengths.values())
                  + list(self.sequence_dense_features.values())):
            v.record_stream(stream)
        for v in list(self.labels.values()) + list(self.sample_weights.values()):
            if v.is_cuda:
                v.record_stream(stream)

    def to_dict(self, sparse_dtype: Optional[torch.dtype] = None) -> Dict[str, torch.Tensor]:
        """Flat feature-tensor dict with the reference's key names (tzrec/datasets/utils.py:465-512):
        `<key>.values` / `.lengths` / `.weights` / `.key_lengths`, labels and sample weights by name."""
        out: Dict[str, torch.Tensor] = {}
        cast = (lambda t: t.to(sparse_dtype)) if sparse_dtype else (lambda t: t)
        for kt in self.dense_features.values():
            for k, v in kt.to_dict().items():
                out[f"{k}.values"] = v
        for kjt in self.sparse_features.values():
            for k, jt in kjt.to_dict().items():
                out[f"{k}.values"], out[f"{k}.lengths"] = cast(jt.values()), cast(jt.lengths())
                if jt.weights_or_none() is not None:
                    out[f"{k}.weights"] = jt.weights_or_none()
        for kjt in self.sequence_mulval_lengths.values():
            for k, jt in kjt.to_dict().items():
                out[f"{k}.key_lengths"], out[f"{k}.lengths"] = cast(jt.values()), cast(jt.lengths())
        for k, jt in self.sequence_dense_features.items():
            out[f"{k}.values"], out[f"{k}.lengths"] = jt.values(), jt.lengths()
        out.update(self.labels)
        out.update(self.sample_weights)
        return out

    def narrow_ids(self) -> "Batch":
        """The host batch with its sparse ids as int32 for the trip to the device (`sparse.WireKeyedJaggedTensor`): 4
        instead of 8 bytes per id across PCIe -- 10.7 instead of 17.5 MB per 65 536-sample Criteo batch -- widened on the
        device behind the copy (`.to(device)`, `GraphTrainPipeline`).  For the dataloader side (the reference builds its
        batches in dataloader workers, /root/reference/tzrec/datasets/utils.py:344-410); ids must be in [0, 2^31)."""
        return Batch(self.dense_features, {k: (v.narrow_ids() if isinstance(v, KeyedJaggedTensor) else v) for k, v in self.sparse_features.items()},
                     self.labels, self.sample_weights, self.sequence_mulval_lengths, self.sequence_dense_features)

    def pin_memory(self) -> "Batch":
        return Batch(
            {k: KeyedTensor(v.keys(), v.length_per_key(), v.values().pin_memory()) for k, v in self.dense_features.items()},
            {k: v.pin_memory() for k, v in self.sparse_features.items()},
            {k: v.pin_memory() for k, v in self.labels.items()},
            {k: v.pin_memory() for k, v in self.sample_weights.items()},
            {k: v.pin_memory() for k, v in self.sequence_mulval_lengths.items()},
            {k: v.pin_memory() for k, v in self.sequence_dense_features.items()},
        )


class EmbeddingGroup(nn.Module):
    """Applies embedding lookups for the pooled feature groups of a model."""

    def __init__(
        self,
        features: Sequence[FeatureSpec],
        feature_groups: Sequence[FeatureGroupSpec],
        wide_embedding_dim: Optional[int] = None,
        device: Optional[torch.device] = None,
        sparse_optimizer: Optional[SparseOptimizerConfig] = None,
        row_layout: str = "interleaved",
        process_group=None,
        plan: Optional[Dict[str, dict]] = None,
        dp_max_rows: int = 65536,
        global_sharding_types: Sequence[str] = (),
        batch_size: int = 1024,
        exchange: str = "exact",
        use_planner: bool = False,
    ) -> None:
        """`exchange`: "exact" | "capacity" ids exchange of the sharded pooled collections (sharding.py).
        `process_group`: shard the tables over its ranks (the seam DistributedModelParallel fills in
        the reference, tzrec/main.py:783-804): pooled tables go to a ShardedEmbeddingBagCollection
        (placement from `plan`, e.g. planner.plan_tables, or the size heuristic), sequence tables to
        ShardedEmbeddingCollections, `zch` tables to the hash-routed sharded map.  Pooled tables of one
        embedding dim with every feature on one table (DLRM, multi_tower_din, mmoe) take the single
        exchange; several dims / a feature on two tables (DeepFM's wide + deep) / `column_wise` entries in
        `plan` take MixedShardedEmbeddingBagCollection (one exchange lane per dim)."""
        super().__init__()
        self._pg, self._plan_in, self._dp_max_rows = process_group, plan, dp_max_rows
        name_to_feature = {f.name: f for f in features}
        configs: "OrderedDict[str, EmbeddingBagConfig]" = OrderedDict()
        zch_blocks: Dict[str, object] = {}
        # table -> sharding types of the feature config that created it; the rest fall back to the global ones
        self._table_sharding_types: Dict[str, List[str]] = {}
        self._global_sharding_types = list(global_sharding_types)
        feat_group_table: Dict[str, Dict[str, str]] = {}
        self._seq_groups = [g for g in feature_groups if g.group_type == "SEQUENCE"]
        feature_groups = [g for g in feature_groups if g.group_type != "SEQUENCE"]
        self._init_sequence_groups(name_to_feature, device, sparse_optimizer, row_layout)
        for g in feature_groups:
            for fname in g.feature_names:
                f = name_to_feature[fname]
                if f.is_sparse:
                    t = f.embedding_name or f"{f.name}_emb"
                    if g.group_type == "WIDE":
                        t += "_wide"
                    if g.embedding_name_suffix:
                        t += "_" + g.embedding_name_suffix
                    feat_group_table.setdefault(fname, {})[g.group_name] = t
        shared = {f: len(set(m.values())) > 1 for f, m in feat_group_table.items()}
        self._group_feature_names: "OrderedDict[str, List[str]]" = OrderedDict()
        self._group_blocks: "OrderedDict[str, List[tuple]]" = OrderedDict()  # (kind, key, dim)
        self._group_dims: Dict[str, "OrderedDict[str, int]"] = {}
        for g in feature_groups:
            is_wide = g.group_type == "WIDE"
            blocks, dims = [], OrderedDict()
            for fname in g.feature_names:
                f = name_to_feature[fname]
                if f.is_sparse:
                    dim = (wide_embedding_dim or 4) if is_wide else f.embedding_dim
                    tname = feat_group_table[fname][g.group_name]
                    cfg = EmbeddingBagConfig(tname, dim, f.num_embeddings, [fname], f.pooling, trainable=f.trainable,
                                             data_type=f.data_type)
                    if f.zch is not None:
                        zch_blocks[tname] = f.zch
                    if getattr(f, "sharding_types", None) and tname not in self._table_sharding_types:
                        self._table_sharding_types[tname] = list(f.sharding_types)
                    if tname in configs:
                        old = configs[tname]
                        if (old.num_embeddings, old.embedding_dim, old.pooling) != (cfg.num_embeddings, dim, cfg.pooling):
                            raise AssertionError(f"there is a mismatch between {cfg} and {old}, can not share embedding.")
                        if fname not in old.feature_names:
                            old.feature_names.append(fname)
                    else:
                        configs[tname] = cfg
                    out_key = f"{fname}@{tname}" if shared[fname] else fname
                    blocks.append(("sparse", out_key, dim))
                    dims[fname] = dim
                else:
                    if is_wide:
                        raise ValueError(f"dense feature [{fname}] should not be configured in wide group.")
                    blocks.append(("dense", fname, f.value_dim))
                    dims[fname] = f.value_dim
            self._group_feature_names[g.group_name] = list(g.feature_names)
            self._group_blocks[g.group_name] = blocks
            self._group_dims[g.group_name] = dims
        self._dense_dims = OrderedDict((f.name, f.value_dim) for f in features if not f.is_sparse)
        self.has_sparse = len(configs) > 0
        # the lookup writes, per group, the contiguous run(s) of sparse blocks; groups that mix dense
        # columns in get one concat on top
        ebc_groups = {g: [k for kind, k, _ in blocks if kind == "sparse"] for g, blocks in self._group_blocks.items()}
        ebc_groups = {g: ks for g, ks in ebc_groups.items() if ks}
        self._sharded_zch = None
        if self.has_sparse and self._pg is not None and self._plan_in is None and not zch_blocks \
                and (self._table_sharding_types or self._global_sharding_types or use_planner):
            # `embedding_constraints` / `global_embedding_constraints` of the config -> the planner picks among the
            # allowed types per table (tzrec/main.py:783-799; per-table constraints win over the global ones)
            import torch.distributed as dist

            from .planner import TableSpec, Topology, plan_tables

            cons = {t: list(self._table_sharding_types.get(t) or self._global_sharding_types) for t in configs}
            kind = sparse_optimizer.kind if sparse_optimizer is not None else "sgd"
            self._plan_in = plan_tables([TableSpec(c.name, c.num_embeddings, c.embedding_dim, list(c.feature_names), optimizer=kind,
                                                   bytes_per_element=2 if str(getattr(c, "data_type", "FP32")).upper() == "FP16" else 4)
                                         for c in configs.values()], Topology(dist.get_world_size(self._pg)), max(int(batch_size), 1),
                                        constraints={t: v for t, v in cons.items() if v})
        if self.has_sparse and self._pg is not None:
            from .sharding import ShardedEmbeddingBagCollection

            if zch_blocks:
                from .zch import ShardedManagedCollisionEmbeddingBagCollection, zch_config_from_msg

                self._sharded_zch = ShardedManagedCollisionEmbeddingBagCollection(
                    list(configs.values()), {t: zch_config_from_msg(z) for t, z in zch_blocks.items()}, device=device,
                    optimizer=sparse_optimizer, groups=ebc_groups, process_group=self._pg, dp_max_rows=self._dp_max_rows)
                self.ebc = self._sharded_zch.sharded
                zch_blocks = {}
            else:
                cfg_list = list(configs.values())
                feats = [f for c in cfg_list for f in c.feature_names]
                mixed = (len({c.embedding_dim for c in cfg_list}) > 1 or len(set(feats)) != len(feats)
                         or any(p.get("sharding_type") in ("column_wise", "table_column_wise", "grid_shard", "table_row_wise")
                                for p in (self._plan_in or {}).values()))
                if mixed:  # DeepFM's wide + deep tables, column-wise tables: one exchange lane per embedding dim
                    from .sharding import MixedShardedEmbeddingBagCollection

                    self.ebc = MixedShardedEmbeddingBagCollection(cfg_list, device=device, optimizer=sparse_optimizer, groups=ebc_groups,
                                                                  row_layout=row_layout, process_group=self._pg,
                                                                  dp_max_rows=self._dp_max_rows, plan=self._plan_in, exchange=exchange)
                else:
                    self.ebc = ShardedEmbeddingBagCollection(cfg_list, device=device, optimizer=sparse_optimizer,
                                                             groups=ebc_groups, row_layout=row_layout, process_group=self._pg,
                                                             dp_max_rows=self._dp_max_rows, plan=self._plan_in, exchange=exchange)
        else:
            self.ebc = EmbeddingBagCollection(list(configs.values()), device=device, optimizer=sparse_optimizer,
                                              groups=ebc_groups, row_layout=row_layout) if self.has_sparse else None
        if self.ebc is not None and self._pg is None:
            # a feature shared by several tables is named feature@table by the EBC only when the
            # feature really feeds >1 table; align our keys with its naming
            self._ebc_groups = {g: [self._ebc_key(k) for k in ks] for g, ks in ebc_groups.items()}
            self.ebc._groups = self._ebc_groups
        # one learning-rate handle for the pooled and the unpooled collections (sparse LR schedulers
        # mutate fused_optimizer.param_groups, tzrec/main.py:877-879)
        for ec in self.ecs.values():
            if self.ebc is not None and ec.fused_optimizer is not None and self.ebc.fused_optimizer is not None:
                ec.fused_optimizer.param_groups = self.ebc.fused_optimizer.param_groups
        # features with a `zch {...}` block: ids go through the managed-collision remap first
        # (the reference keeps them in a second collection, embedding.py:856-864; here the remap passes
        # the other keys through, so one collection serves both)
        self.mc = None
        if zch_blocks and self.ebc is not None:
            from .zch import ManagedCollisionEmbeddingBagCollection, zch_config_from_msg

            self.mc = ManagedCollisionEmbeddingBagCollection(self.ebc, {t: zch_config_from_msg(z) for t, z in zch_blocks.items()})

    # -- SEQUENCE groups (SequenceEmbeddingGroupImpl, tzrec/modules/embedding.py:993-1498) ---------------
    # Every sparse feature of a sequence group is looked up UNPOOLED through its own tables (an
    # EmbeddingCollection per embedding dim, embedding.py:1193-1197; tables are NOT shared with the
    # pooled collection even when a feature also sits in a DEEP group).  Outputs per group g:
    #   g.query            [B, sum D_q]      non-sequence features: the single id's row (raw features: values)
    #   g.sequence         [B, Lmax, sum D_s] sequence features padded to the longest sequence of the batch
    #   g.sequence_length  [B]               lengths of the group's first sequence feature
    def _init_sequence_groups(self, name_to_feature, device, sparse_optimizer, row_layout) -> None:
        from .sequence import EmbeddingCollection, EmbeddingConfig

        self._seq_info: "OrderedDict[str, dict]" = OrderedDict()
        # pad sequence groups to their configured `sequence_length` (no device read-back of the batch's longest sequence: see
        # forward); off by default = the reference's shapes (tzrec/modules/embedding.py:1431-1446)
        self.static_sequence_padding = False
        # sequence groups handed to their encoders as ROWS -- `<g>.sequence_jagged` [N, sum D_s], `<g>.sequence_offsets`
        # [B + 1], `<g>.sequence_max_len` -- instead of the padded `<g>.sequence` [B, L, sum D_s]: set by the models whose
        # encoders evaluate the jagged form (sequence.DINEncoder.forward_jagged); no padding position is computed
        self.jagged_sequence_groups: set = set()
        by_dim: Dict[int, "OrderedDict[str, EmbeddingConfig]"] = {}
        seq_constraints: Dict[str, str] = {}
        for g in self._seq_groups:
            q, sq = [], []
            for fname in g.feature_names:
                f = name_to_feature[fname]
                (sq if f.is_sequence else q).append(f)
                if f.is_sparse:
                    t = (f.embedding_name or f"{f.name}_emb") + (f"_{g.embedding_name_suffix}" if g.embedding_name_suffix else "")
                    cfgs = by_dim.setdefault(f.embedding_dim, OrderedDict())
                    if t in cfgs:
                        if cfgs[t].num_embeddings != f.num_embeddings:
                            raise AssertionError(f"there is a mismatch between tables named {t}, can not share embedding.")
                        if fname not in cfgs[t].feature_names:
                            cfgs[t].feature_names.append(fname)
                    else:
                        cfgs[t] = EmbeddingConfig(t, f.embedding_dim, f.num_embeddings, [fname])
                        allowed = list(getattr(f, "sharding_types", None) or self._global_sharding_types)
                        if allowed:  # the unpooled exchange places a table row-wise (default) or table-wise
                            ok = [k for k in allowed if k in ("row_wise", "table_wise", "table_row_wise")]
                            if not ok:
                                raise ValueError(f"sequence table {t}: sharding types {allowed} are not executable for unpooled "
                                                 "lookups (row_wise or table_wise)")
                            if "row_wise" not in ok and "table_row_wise" not in ok:
                                seq_constraints[t] = "table_wise"
            if not sq:
                raise ValueError(f"sequence group {g.group_name} has no sequence feature")
            self._seq_info[g.group_name] = {
                "query": q, "sequence": sq, "max_len": max(f.sequence_length for f in sq) or 0,
                "query_dim": sum(f.embedding_dim if f.is_sparse else f.value_dim for f in q),
                "sequence_dim": sum(f.embedding_dim if f.is_sparse else f.value_dim for f in sq)}
        if self._pg is not None:
            from .sequence import ShardedEmbeddingCollection

            self.ecs = nn.ModuleDict({str(d): ShardedEmbeddingCollection(
                list(c.values()), device=device, optimizer=sparse_optimizer, process_group=self._pg,
                constraints={t: k for t, k in seq_constraints.items() if t in c} or None) for d, c in by_dim.items()})
        else:
            self.ecs = nn.ModuleDict({str(d): EmbeddingCollection(list(c.values()), device=device, optimizer=sparse_optimizer,
                                                                  row_layout=row_layout) for d, c in by_dim.items()})
        self._ec_keys = {str(d): [f for c in cfgs.values() for f in c.feature_names] for d, cfgs in by_dim.items()}
        seen: Dict[str, FeatureSpec] = {}
        for info in self._seq_info.values():
            for f in info["sequence"]:
                if f.is_sparse and f.value_dim != 1:
                    seen.setdefault(f.name, f)
        self._seq_mulval = list(seen.values())

    def _forward_sequence_groups(self, sparse: KeyedJaggedTensor, dense_cols: Dict[str, torch.Tensor],
                                 mulval: Optional[KeyedJaggedTensor] = None,
                                 seq_dense: Optional[Dict[str, object]] = None) -> Dict[str, torch.Tensor]:
        from .sequence import JaggedTensor, jagged_to_padded_dense, segment_reduce

        jts = {}
        for info in self._seq_info.values():  # raw sub-features of a sequence: values travel in the batch
            for f in info["sequence"]:
                if not f.is_sparse:
                    if not seq_dense or f.name not in seq_dense:
                        raise KeyError(f"sequence feature {f.name} is dense: the batch must carry it in `sequence_dense_features`")
                    jts[f.name] = seq_dense[f.name]
        index = {k: i for i, k in enumerate(sparse.keys())}
        for d, ec in self.ecs.items():
            keys = self._ec_keys[d]
            jts.update(ec(sparse.permute([index[k] for k in keys])))
        # multi-valued steps: the rows of a step's ids are pooled into one row per step
        # (SequenceEmbeddingGroupImpl, tzrec/modules/embedding.py:1353-1366)
        for f in self._seq_mulval:
            if mulval is None or f.name not in mulval.keys():
                raise KeyError(f"sequence feature {f.name} has value_dim {f.value_dim}: the batch must carry its "
                               "per-step id counts in `sequence_mulval_lengths`")
            i = mulval.keys().index(f.name)
            Bm = mulval.stride()
            lo, hi = int(mulval.offsets()[i * Bm]), int(mulval.offsets()[(i + 1) * Bm])
            key_lengths = mulval.values()[lo:hi]
            seq_lengths = mulval.lengths()[i * Bm:(i + 1) * Bm]
            off = torch.zeros(Bm + 1, dtype=torch.int64, device=seq_lengths.device)
            torch.cumsum(seq_lengths.to(torch.int64), 0, out=off[1:])
            jts[f.name] = JaggedTensor(segment_reduce(jts[f.name].values(), key_lengths, f.pooling), seq_lengths, off)
        out: Dict[str, torch.Tensor] = {}
        for g, info in self._seq_info.items():
            qs = []
            for f in info["query"]:
                if f.is_sparse:  # single id per sample: its row (to_padded_dense(1).squeeze(1), embedding.py:1428-1430)
                    jt = jts[f.name]
                    qs.append(jagged_to_padded_dense(jt.values(), jt.offsets(), 1).squeeze(1))
                else:
                    qs.append(dense_cols[f.name])
            if qs:
                out[f"{g}.query"] = torch.cat(qs, dim=1)
            first = jts[info["sequence"][0].name]
            lens = first.lengths().to(torch.int64)
            if g in self.jagged_sequence_groups:
                out[f"{g}.sequence_length"] = lens
                out[f"{g}.sequence_jagged"] = torch.cat([jts[f.name].values() for f in info["sequence"]], dim=1)
                out[f"{g}.sequence_offsets"] = first.offsets()
                out[f"{g}.sequence_max_len"] = info["max_len"]  # (the padded length the reference uses; rank_model adds a group here only when it is configured)
                continue
            if self.static_sequence_padding and info["max_len"]:
                # pad to the configured sequence_length instead of the batch's longest sequence: no host sync (the step can be
                # captured in a hipGraph) and one shape for every batch; the positions behind a sample's length are masked by
                # the sequence encoders either way (`sequence_length` goes with the tensor)
                lmax = info["max_len"]
            else:
                if lens.is_cuda and torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("the padded length of a sequence group is the batch's longest sequence, read back from the "
                                       "device: not capturable -- set EmbeddingGroup.static_sequence_padding = True (needs "
                                       "`sequence_length` in the feature config)")
                lmax = max(int(lens.max().item()) if lens.numel() else 0, 1)  # one host sync, as the reference's fx_int_item
                if info["max_len"]:
                    lmax = min(lmax, info["max_len"])
            out[f"{g}.sequence_length"] = lens
            out[f"{g}.sequence"] = torch.cat(
                [jagged_to_padded_dense(jts[f.name].values(), jts[f.name].offsets(), lmax) for f in info["sequence"]], dim=-1)
        return out

    def _ebc_key(self, out_key: str) -> str:
        return out_key if out_key in self.ebc._out_dim else out_key.split("@")[0]

    # -- introspection used by the models (embedding.py:886-907) -----------------------------
    def parameter_constraints(self, prefix: str = "") -> Dict[str, Dict[str, List[str]]]:
        """{`<prefix>ebc.<table>`: {"sharding_types": [...]}} for the tables whose feature config carries
        `embedding_constraints` (EmbeddingGroupImpl.parameter_constraints, tzrec/modules/embedding.py:898-907)."""
        return {f"{prefix}ebc.{t}": {"sharding_types": list(v)} for t, v in self._table_sharding_types.items()}

    def group_names(self) -> List[str]:
        return list(self._group_feature_names)

    def has_group(self, name: str) -> bool:
        return name in self._group_feature_names

    def group_dims(self, name: str) -> List[int]:
        return list(self._group_dims[name].values())

    def group_feature_dims(self, name: str) -> Dict[str, int]:
        return self._group_dims[name]

    def group_total_dim(self, name: str) -> int:
        if "." in name:  # "<seq group>.query" / "<seq group>.sequence"
            g, part = name.rsplit(".", 1)
            return self._seq_info[g][f"{part}_dim"]
        return sum(self._group_dims[name].values())

    @property
    def fused_optimizer(self):
        if self.ebc is not None:
            return self.ebc.fused_optimizer
        for ec in self.ecs.values():
            return ec.fused_optimizer
        return None

    def forward(self, batch: Batch) -> Dict[str, torch.Tensor]:
        sparse = batch.sparse_features.get(BASE_DATA_GROUP)
        dense = batch.dense_features.get(BASE_DATA_GROUP)
        if self.mc is not None:
            self.mc.train(self.training)
            sparse = self.mc.remap_step(sparse)
        if self._sharded_zch is not None:  # owner-side remap / eviction happen inside the sharded exchange
            self._sharded_zch.train(self.training)
            pooled = self._sharded_zch.forward_grouped(sparse)
        else:
            pooled = self.ebc.forward_grouped(sparse) if self.ebc is not None else {}
        if self.mc is not None:
            self.mc.finish_step()
        dense_cols = dense.to_dict() if dense is not None else {}
        out: Dict[str, torch.Tensor] = {}
        for g, blocks in self._group_blocks.items():
            if all(kind == "sparse" for kind, _, _ in blocks):
                out[g] = pooled[g]
                continue
            parts, col, run = [], 0, None
            # pooled[g] holds the group's sparse blocks back to back, in order
            for kind, key, dim in blocks:
                if kind == "sparse":
                    run = (run[0], run[1] + dim) if run else (col, col + dim)
                    col += dim
                else:
                    if run:
                        parts.append(pooled[g][:, run[0]:run[1]])
                        run = None
                    parts.append(dense_cols[key])
            if run:
                parts.append(pooled[g][:, run[0]:run[1]])
            out[g] = torch.cat(parts, dim=1)
        if self._seq_info:
            out.update(self._forward_sequence_groups(batch.sparse_features.get(BASE_DATA_GROUP), dense_cols,
                                                     batch.sequence_mulval_lengths.get(BASE_DATA_GROUP),
                                                     batch.sequence_dense_features))
        return out


def _losses_and_predictions(model, loss_fn, batch):
    """The training forward of the reference's TrainWrapper (/root/reference/tzrec/models/model.py:271-297) is
    predictions AND losses in one call; a model that offers `loss_and_predictions(batch) -> (losses, predictions)` gets
    to compute them together (DLRM: the top MLP's tail, the loss and their backward in one launch), any other model
    is called for its predictions and `loss_fn(predictions, batch)` follows."""
    fused = getattr(model, "loss_and_predictions", None)
    if fused is not None:
        return fused(batch)
    predictions = model(batch)
    return loss_fn(predictions, batch), predictions


def _backward_of_losses(losses) -> None:
    """backward of the unweighted sum of the named losses (TrainWrapper.forward, tzrec/models/model.py:293): every loss
    receives autograd's 1.0, which `dense.root_loss` lets the fused loss kernels skip multiplying by."""
    from .dense import root_loss, unit_gradient

    vals = list(losses.values())
    total = vals[0] if len(vals) == 1 else sum(vals[1:], vals[0])
    with root_loss():
        total.backward(gradient=unit_gradient(total))


class TrainPipeline:
    """Minimal ``pipeline.progress(iterator)`` (tzrec/utils/dist_util.py:221-303,336-377 ->
    torchrec TrainPipelineSparseDist [upstream]): the next batch is copied host->device on a memcpy
    stream while the current one computes; the backward index plan of the embedding lookup runs on
    its own stream (the analogue of the data-dist stage); the sparse update happens inside
    ``loss.backward()``; ``optimizer.step()`` only touches the dense parameters."""

    def __init__(self, model: nn.Module, optimizer: torch.optim.Optimizer, device: torch.device, loss_fn,
                 fetch_first: bool = True) -> None:
        self._model, self._opt, self._device, self._loss_fn = model, optimizer, torch.device(device), loss_fn
        self._fetch_first = bool(fetch_first)
        self._copy_stream = torch.cuda.Stream(device=self._device) if self._device.type == "cuda" else None
        self._next: Optional[Batch] = None
        self._exhausted = False

    def _fetch(self, it) -> Optional[Batch]:
        try:
            b = next(it)
        except StopIteration:
            self._exhausted = True
            return None
        if self._copy_stream is None:
            return b.to(self._device)
        with torch.cuda.stream(self._copy_stream):
            return b.to(self._device, non_blocking=True)

    def progress(self, dataloader_iter):
        if self._next is None and not self._exhausted:
            self._next = self._fetch(dataloader_iter)
        if self._next is None:
            raise StopIteration
        batch = self._next
        if self._copy_stream is not None:
            torch.cuda.current_stream(self._device).wait_stream(self._copy_stream)
            batch.record_stream(torch.cuda.current_stream(self._device))
        if self._fetch_first:
            self._next = self._fetch(dataloader_iter)  # overlaps with the step below
        self._opt.zero_grad(set_to_none=True)
        losses, predictions = _losses_and_predictions(self._model, self._loss_fn, batch)
        _backward_of_losses(losses)
        if hasattr(self._model, "allreduce_dense_grads"):
            self._model.allreduce_dense_grads()  # no-op unless the model was built over a process group
        self._opt.step()
        if not self._fetch_first:
            # queued behind the step's launches (what GraphTrainPipeline does: one launch, copy right behind it).  For
            # this eager pipeline it is the slower order (1.29 vs 1.13 ms at B = 65 536, profiles/r02u: the copy only
            # starts once the host has queued the ~60 launches of the step), hence fetch_first = True by default
            self._next = self._fetch(dataloader_iter)
        return losses, predictions, batch


def _batch_tensors(b: Batch, skip_constant: bool = True):
    """Every tensor of a batch that changes from batch to batch, in a fixed order (for slot-to-slot copies of
    same-shape batches)."""
    out = []
    for g in sorted(b.dense_features):
        out.append(b.dense_features[g].values())
    for g in sorted(b.sparse_features):
        k = b.sparse_features[g]
        # one id per bag (Criteo): the lengths are all ones and the offsets unused -- constant across batches, so
        # they stay in the slot and never cross PCIe again (6.8 of 24.4 MB per step at B = 65 536)
        const_lengths = skip_constant and k.uniform_length() == 1
        vals = k.wire_values() if hasattr(k, "wire_values") else k.values()  # (a host batch with int32 ids on the wire)
        out += [t for t in (vals, None if const_lengths else k.lengths_or_none(), k.weights_or_none(),
                            None if const_lengths else k.offsets_or_none()) if t is not None]
    for n in sorted(b.labels):
        out.append(b.labels[n])
    for n in sorted(b.sample_weights):
        out.append(b.sample_weights[n])
    return out


def zch_wrapper_of(model: nn.Module):
    """the model's zero-collision-hash wrapper (`embedding_group.mc`), or None"""
    eg = getattr(model, "embedding_group", None)
    if eg is None and hasattr(model, "m"):  # (a thin wrapper around the model, as bench.py's)
        eg = getattr(model.m, "embedding_group", None)
    return getattr(eg, "mc", None) if eg is not None else None


def after_graph_replay(model: nn.Module) -> None:
    """Host bookkeeping behind one replay of a captured training step: a zero-collision hash in ring mode counts the step and
    runs the admission / eviction round that falls due (zch.ManagedCollisionEmbeddingBagCollection.replayed)."""
    mc = zch_wrapper_of(model)
    if mc is not None and getattr(mc, "device_profile", False):
        mc.replayed()


class GraphTrainPipeline:
    """`pipeline.progress(iterator)` with the whole step replayed from a hipGraph while the NEXT batch
    crosses PCIe (tzrec/utils/dist_util.py:221-303: H2D of batch i+1 on the memcpy stream under the
    forward / backward of batch i).  For fixed-shape batches (Criteo: one id per bag): two device slots;
    batch i+1 is copied from pinned host memory into the idle slot on the copy stream, and the step of a
    slot is captured once (forward, loss, backward with the fused sparse update, dense optimizer) and
    replayed.  Batches whose shapes differ from the slots' fall back to the eager `TrainPipeline` step.

    The model must be capturable (dense optimizer with device-side step counts: `dense.FusedDenseAdam`,
    or torch optimizers built with `capturable=True`)."""

    def __init__(self, model: nn.Module, optimizer, device: torch.device, loss_fn, warmup: int = 2,
                 stage_first: bool = False) -> None:
        """`stage_first`: queue the next batch's H2D before (True) or after (False, default) this step's launch.
        Queued first, the copy did not overlap the replay at all (1.21 ms per step at B = 65 536 = step + copy);
        queued after the launch it hides under it (0.88 ms; profiles/r02u)."""
        self._stage_first = bool(stage_first)
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # a sharded model's step contains RCCL calls; captured into a hipGraph they took the process down on this
            # stack (profiles/r02p) -- sharded_step.ShardedTrainStep(step_graph=True) cuts the step at its collectives
            raise RuntimeError("GraphTrainPipeline captures the whole step in one hipGraph: single-process models only "
                               "(sharded models: sharded_step.ShardedTrainStep)")
        self._model, self._opt, self._device, self._loss_fn = model, optimizer, torch.device(device), loss_fn
        assert self._device.type == "cuda", "GraphTrainPipeline replays hipGraphs: CUDA/HIP device only"
        mc = zch_wrapper_of(model)
        if mc is not None:  # the step of a model with a zero-collision hash is capturable in ring mode (zch.py)
            mc.device_profile = True
        self._copy_stream = torch.cuda.Stream(device=self._device)
        from .dense import unit_gradient
        unit_gradient(torch.zeros((), dtype=torch.float32, device=self._device))  # (made before any capture, see dense.unit_gradient)
        self._slots = [None, None]     # device batches with static addresses
        self._graphs = [None, None]    # (CUDAGraph, losses, predictions) per slot
        self._seen = [0, 0]
        self._ready = [None, None]     # event: the H2D copy into the slot has finished
        self._done = [None, None]      # event: the last step that read the slot has finished
        self._warmup, self._pool = warmup, None
        self._pending = None           # (slot, host batch) of the batch copied ahead
        self._wire = {}                # (slot, tensor index) -> int32 staging buffer of ids that cross PCIe narrowed
        self._widen = [[], []]         # per slot: (int64 ids of the slot, their int32 staging buffer) still to be widened
        self._exhausted, self._i = False, 0

    def _capturable(self, batch) -> bool:
        """a model with a zero-collision hash replays from a graph in ring mode, which needs uniform bags (zch.py): slots that
        hold jagged id lists keep stepping eagerly instead of failing inside a capture"""
        if zch_wrapper_of(self._model) is None:
            return True
        return all(k.uniform_length() for k in getattr(batch, "sparse_features", {}).values())

    def _stage(self, it):
        """next host batch -> the idle slot, on the copy stream"""
        try:
            hb = next(it)
        except StopIteration:
            self._exhausted = True
            return None
        slot = self._i % 2
        self._i += 1
        with torch.cuda.stream(self._copy_stream):
            if self._slots[slot] is None:
                self._slots[slot] = hb.to(self._device, non_blocking=True)
            else:
                if self._done[slot] is not None:
                    self._copy_stream.wait_event(self._done[slot])  # the step that still reads this slot
                src, dst = _batch_tensors(hb), _batch_tensors(self._slots[slot])
                narrow = lambda a, b: a.dtype == torch.int32 and b.dtype == torch.int64  # noqa: E731 -- ids on the int32 wire
                if len(src) != len(dst) or any(a.shape != b.shape or (a.dtype != b.dtype and not narrow(a, b)) for a, b in zip(src, dst)):
                    raise ValueError("GraphTrainPipeline needs fixed-shape batches (shapes changed between batches)")
                widen = []
                for j, (a, b) in enumerate(zip(src, dst)):
                    if a.dtype == b.dtype:
                        b.copy_(a, non_blocking=True)
                    else:  # int32 over PCIe into a staging buffer of the slot; widened into the slot's int64 ids on the STEP's
                        # stream, in front of the step (progress): as a kernel of the copy stream it ran NEXT TO the previous
                        # step's kernels -- 11-56 us of a second queue's kernel per step, and two queues share the chip badly
                        # here (profiles/r05as: e2e 0.595 ms against 0.536 resident)
                        stg = self._wire.setdefault((slot, j), torch.empty(a.shape, dtype=torch.int32, device=self._device))
                        stg.copy_(a, non_blocking=True)
                        widen.append((b, stg))
                self._widen[slot] = widen
            ev = torch.cuda.Event()
            ev.record()
            self._ready[slot] = ev
        return slot, hb

    def _step(self, batch: Batch):
        self._opt.zero_grad(set_to_none=True)
        losses, predictions = _losses_and_predictions(self._model, self._loss_fn, batch)
        _backward_of_losses(losses)
        self._opt.step()
        return losses, predictions

    def progress(self, dataloader_iter):
        if self._pending is None and not self._exhausted:
            self._pending = self._stage(dataloader_iter)
        if self._pending is None:
            raise StopIteration
        slot, _ = self._pending
        cur = torch.cuda.current_stream(self._device)
        cur.wait_event(self._ready[slot])
        batch = self._slots[slot]
        for dst_ids, stg in self._widen[slot]:  # (see _stage)
            dst_ids.copy_(stg)
        self._widen[slot] = []
        if self._stage_first:
            self._pending = self._stage(dataloader_iter)  # batch i+1 crosses PCIe under the step below
        from .dense import lr_sync_targets, sync_learning_rates

        if getattr(self, "_lr_targets", None) is None:
            self._lr_targets = lr_sync_targets(self._model, self._opt)
        sync_learning_rates(self._model, self._opt, self._lr_targets)  # outside capture: the graphs read the rates from device scalars
        if self._graphs[slot] is None and self._seen[slot] >= self._warmup and self._capturable(batch):
            g = torch.cuda.CUDAGraph()
            # capture on the caller's stream when it is a side stream (autograd's accumulation nodes and the
            # captured kernels then agree on it); torch picks one when the caller sits on the default stream
            kw = {} if cur == torch.cuda.default_stream(self._device) else {"stream": cur}
            with torch.cuda.graph(g, pool=self._pool, **kw):
                losses, predictions = self._step(batch)
            self._pool = g.pool()
            self._graphs[slot] = (g, losses, predictions)
        if self._graphs[slot] is not None:
            g, losses, predictions = self._graphs[slot]
            g.replay()
            after_graph_replay(self._model)
        else:
            losses, predictions = self._step(batch)
            self._seen[slot] += 1
        ev = torch.cuda.Event()
        ev.record(cur)
        self._done[slot] = ev
        if not self._stage_first:
            self._pending = self._stage(dataloader_iter)  # ... queued behind the launch of the step above
        return losses, predictions, batch

