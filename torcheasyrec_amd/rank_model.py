"""Config-driven DLRM / DeepFM on top of EmbeddingGroup (the reference's model API surface).

`build_rank_model(spec)` takes what `config.load_pipeline_spec` read from a tzrec pipeline config
and builds the model the reference would build from the same file
(/root/reference/tzrec/main.py:134-166 picks the class from the model_config oneof;
/root/reference/tzrec/models/dlrm.py:26-135, deepfm.py:26-108, rank_model.py:83-262).
`forward(batch) -> {"logits", "probs"}`; `loss(predictions, batch) -> {"binary_cross_entropy": ...}`.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from .config import PipelineSpec
from .dlrm import MLP, OutputLinear, _on_emulator
from .embedding import SparseOptimizerConfig
from .embedding_group import Batch, EmbeddingGroup
from .interaction import FactorizationMachine, dot_interaction


def mlp_kwargs(msg) -> Dict[str, object]:
    """An MLP proto block (tzrec/protos/module.proto:4-17: hidden_units, dropout_ratio, activation,
    use_bn, bias, use_ln) as the module's keyword arguments (the reference's config_to_kwargs)."""
    return {"hidden_units": [int(x) for x in msg.many("hidden_units")], "bias": bool(msg.one("bias", True)),
            "activation": str(msg.one("activation", "nn.ReLU")), "use_bn": bool(msg.one("use_bn", False)),
            "dropout_ratio": [float(x) for x in msg.many("dropout_ratio")], "use_ln": bool(msg.one("use_ln", False))}


def mlp_from_msg(in_features: int, msg) -> MLP:
    """Every tower of the config-built models goes through here: no MLP field is dropped silently."""
    return MLP(in_features, **mlp_kwargs(msg))


class RankModel(nn.Module):
    # set by build_rank_model(..., process_group=...): tables sharded over the group's ranks, dense
    # parameters data-parallel (the DistributedModelParallel seam, tzrec/main.py:783-804)
    _build_pg = None
    _build_plan = None
    _build_use_planner = False
    _build_exchange = "exact"

    def __init__(self, spec: PipelineSpec, device: Optional[torch.device], sparse_optimizer: Optional[SparseOptimizerConfig]) -> None:
        super().__init__()
        self._spec = spec
        self._label = spec.label_fields[0] if spec.label_fields else "label"
        self._pg = RankModel._build_pg
        self.embedding_group = EmbeddingGroup(
            spec.features, spec.feature_groups, wide_embedding_dim=spec.wide_embedding_dim or None,
            device=device, sparse_optimizer=sparse_optimizer if sparse_optimizer is not None else spec.sparse_optimizer,
            process_group=self._pg, plan=RankModel._build_plan,
            global_sharding_types=getattr(spec, "global_sharding_types", ()), batch_size=spec.batch_size or 1024,
            use_planner=RankModel._build_use_planner, exchange=RankModel._build_exchange)

    def sync_dense_parameters(self) -> None:
        """Same dense parameters on every rank (DDP broadcasts rank 0's at construction)."""
        if self._pg is not None:
            import torch.distributed as dist

            from .sharding import stream_collective

            for p in self.dense_parameters():
                stream_collective(dist.broadcast, p.data, src=0, group=self._pg)

    def allreduce_dense_grads(self) -> None:
        """DDP semantics for the dense parameters: average their gradients over the ranks (one flat
        all-reduce); sparse gradients were already applied by the owners inside backward."""
        if self._pg is None:
            return
        from .sharding import allreduce_average

        allreduce_average([p.grad for p in self.dense_parameters() if p.grad is not None], self._pg)

    def build_input(self, batch: Batch) -> Dict[str, torch.Tensor]:
        return self.embedding_group(batch)

    def dense_parameters(self):
        for n, p in self.named_parameters():
            if not n.startswith("embedding_group."):
                yield p

    @property
    def fused_optimizer(self):
        return self.embedding_group.fused_optimizer

    def _output_to_prediction(self, y: torch.Tensor) -> Dict[str, torch.Tensor]:
        logits = torch.squeeze(y, dim=1)  # rank_model.py:142-146 (num_class == 1)
        return {"logits": logits, "probs": torch.sigmoid(logits)}

    def loss(self, predictions: Dict[str, torch.Tensor], batch: Batch) -> Dict[str, torch.Tensor]:
        label = batch.labels[self._label].to(torch.float32)
        return {"binary_cross_entropy": nn.functional.binary_cross_entropy_with_logits(predictions["logits"], label)}


class ConfigDLRM(RankModel):
    def __init__(self, spec: PipelineSpec, device=None, sparse_optimizer=None) -> None:
        super().__init__(spec, device, sparse_optimizer)
        eg, m = self.embedding_group, spec.model
        names = eg.group_names()
        self._sparse_group = names[0] if len(names) == 1 else "sparse"
        self._dense_group = "dense"
        self.dense_mlp = None
        if len(names) > 1 and eg.has_group(self._dense_group):
            self.dense_mlp = mlp_from_msg(eg.group_total_dim(self._dense_group), m.one("dense_mlp"))
        dims = set(eg.group_dims(self._sparse_group))
        if len(dims) > 1:
            raise Exception(f"sparse group feature dims must be the same, but we find {dims}")
        self._dim = dims.pop()
        self._num_sparse = len(eg.group_dims(self._sparse_group))
        if self.dense_mlp and self._dim != self.dense_mlp.output_dim():
            raise Exception("dense mlp last hidden_unit must be the same sparse feature dim")
        self._arch_with_sparse = bool(m.one("arch_with_sparse", True))
        n = self._num_sparse + (1 if self.dense_mlp else 0)
        feat = n * (n - 1) // 2 + (self._dim if self.dense_mlp else 0) + (self._num_sparse * self._dim if self._arch_with_sparse else 0)
        self.final_mlp = mlp_from_msg(feat, m.one("final"))
        self.output_mlp = OutputLinear(self.final_mlp.output_dim(), spec.num_class)
        if device is not None:
            for mod in (self.dense_mlp, self.final_mlp, self.output_mlp):
                if mod is not None:
                    mod.to(device)

    def forward(self, batch: Batch) -> Dict[str, torch.Tensor]:
        g = self.build_input(batch)
        sparse = g[self._sparse_group]
        d = self.dense_mlp(g[self._dense_group]) if self.dense_mlp else None
        allf = dot_interaction(d, sparse, self._dim, cat_dense=True, cat_sparse=self._arch_with_sparse)
        return self._output_to_prediction(self.output_mlp(self.final_mlp(allf)))


class ConfigDeepFM(RankModel):
    def __init__(self, spec: PipelineSpec, device=None, sparse_optimizer=None) -> None:
        super().__init__(spec, device, sparse_optimizer)
        eg, m = self.embedding_group, spec.model
        self._has_fm = eg.has_group("fm")
        fm_dims = eg.group_dims("fm" if self._has_fm else "deep")
        assert len(set(fm_dims)) == 1, f"embedding dimension of fm features must be same. but got {set(fm_dims)}"
        self._fm_n, self._fm_dim = len(fm_dims), fm_dims[0]
        self.fm = FactorizationMachine()
        self.deep_mlp = mlp_from_msg(eg.group_total_dim("deep"), m.one("deep"))
        final_dim = self.deep_mlp.output_dim()
        self.final_mlp = None
        if m.has("final"):
            self.final_mlp = mlp_from_msg(1 + self._fm_dim + final_dim, m.one("final"))
            final_dim = self.final_mlp.output_dim()
        self.output_mlp = OutputLinear(final_dim, spec.num_class)
        if device is not None:
            for mod in (self.deep_mlp, self.final_mlp, self.output_mlp):
                if mod is not None:
                    mod.to(device)

    def forward(self, batch: Batch) -> Dict[str, torch.Tensor]:
        g = self.build_input(batch)
        y_wide = torch.sum(g["wide"], dim=1, keepdim=True)
        y_deep = self.deep_mlp(g["deep"])
        fm_feat = (g["fm"] if self._has_fm else g["deep"]).reshape(-1, self._fm_n, self._fm_dim)
        y_fm = self.fm(fm_feat)
        if self.final_mlp is not None:
            y = self.output_mlp(self.final_mlp(torch.cat([y_wide, y_fm, y_deep], dim=1)))
        else:
            y = y_wide + torch.sum(y_fm, dim=1, keepdim=True) + self.output_mlp(y_deep)
        return self._output_to_prediction(y)


JAGGED_DIN = True  # multi_tower_din: DIN towers on the jagged positions (False: the reference's padded tensors)


class ConfigMultiTowerDIN(RankModel):
    """`multi_tower_din {...}` (tzrec/models/multi_tower_din.py:36-111): one MLP tower per DEEP group,
    one DIN target-attention tower per SEQUENCE group, concatenated, optional final MLP, logits layer."""

    def __init__(self, spec: PipelineSpec, device=None, sparse_optimizer=None) -> None:
        super().__init__(spec, device, sparse_optimizer)
        from .sequence import DIN_JAGGED_MAX_LEN, DINEncoder

        eg, m = self.embedding_group, spec.model
        self.towers = nn.ModuleDict()
        total = 0
        for tower in m.many("towers"):
            g = str(tower.one("input"))
            mlp = mlp_from_msg(eg.group_total_dim(g), tower.one("mlp"))
            self.towers[g] = mlp
            total += mlp.output_dim()
        self.din_towers = nn.ModuleList()
        for tower in m.many("din_towers"):
            g = str(tower.one("input"))
            din = DINEncoder(eg.group_total_dim(f"{g}.sequence"), eg.group_total_dim(f"{g}.query"), g,
                             attn_mlp={"hidden_units": [int(x) for x in tower.one("attn_mlp").many("hidden_units")]})
            self.din_towers.append(din)
            total += din.output_dim()
            # the tower takes the group's sequences as rows of the unpooled lookup (no padding) when its attention MLP is
            # the plain Linear + ReLU stack and the group's sequence features come from ONE sequence_feature block (one
            # set of lengths): csrc/din_attention.hip
            info = eg._seq_info.get(g) if hasattr(eg, "_seq_info") else None
            if JAGGED_DIN and info is not None and din.jagged_capable() and len({f.name.split("__")[0] for f in info["sequence"]}) == 1 \
                    and not any(getattr(f, "value_dim", 1) not in (0, 1) for f in info["sequence"]) \
                    and info.get("max_len") and int(info["max_len"]) <= DIN_JAGGED_MAX_LEN:
                # (a group without a configured sequence_length pads to the batch's longest sequence in the reference: nothing
                # is truncated there, and the jagged kernels keep at most DIN_JAGGED_MAX_LEN positions of a sample -- such
                # groups stay on the padded form)
                eg.jagged_sequence_groups.add(g)
        self.final_mlp = None
        if m.has("final"):
            self.final_mlp = mlp_from_msg(total, m.one("final"))
            total = self.final_mlp.output_dim()
        self.output_mlp = OutputLinear(total, spec.num_class)
        if device is not None:
            for mod in (self.towers, self.din_towers, self.final_mlp, self.output_mlp):
                if mod is not None:
                    mod.to(device)

    def forward(self, batch: Batch) -> Dict[str, torch.Tensor]:
        g = self.build_input(batch)
        outs = [mlp(g[name]) for name, mlp in self.towers.items()]
        outs += [din(g) for din in self.din_towers]
        y = torch.cat(outs, dim=-1)
        if self.final_mlp is not None:
            y = self.final_mlp(y)
        return self._output_to_prediction(self.output_mlp(y))


FUSED_MOE_MIX = True  # A/B switch: False = the reference's stack / softmax / batched-matmul form


class MMoE(nn.Module):
    """Multi-gate mixture of experts, the reference module's constructor and forward
    (tzrec/modules/mmoe.py:20-76): `num_expert` expert MLPs over the input, per task an optional
    gate MLP and a Linear + softmax over the experts; returns one mixed tensor per task."""

    def __init__(self, in_features: int, expert_mlp: Dict[str, object], num_expert: int, num_task: int,
                 gate_mlp: Optional[Dict[str, object]] = None) -> None:
        super().__init__()
        self.num_expert, self.num_task = num_expert, num_task
        self.expert_mlps = nn.ModuleList([MLP(in_features, **expert_mlp) for _ in range(num_expert)])
        gate_in = in_features
        self.has_gate_mlp = gate_mlp is not None
        if self.has_gate_mlp:
            self.gate_mlps = nn.ModuleList([MLP(in_features, **gate_mlp) for _ in range(num_task)])
            gate_in = self.gate_mlps[0].hidden_units[-1]
        self.gate_finals = nn.ModuleList([OutputLinear(gate_in, num_expert) for _ in range(num_task)])  # (nn.Linear's parameters; <= 8 units: one-pass kernels)

    def output_dim(self) -> int:
        return self.expert_mlps[0].hidden_units[-1]

    def forward(self, input: torch.Tensor):
        outs = [e(input) for e in self.expert_mlps]
        logits = [self.gate_finals[i](self.gate_mlps[i](input) if self.has_gate_mlp else input) for i in range(self.num_task)]
        if FUSED_MOE_MIX and input.dim() == 2 and (input.is_cuda or _on_emulator()):
            from .dense import moe_mix, moe_mix_ok

            if moe_mix_ok(logits, outs):  # softmax + mixing of every task in one launch per direction, nothing stacked (csrc/moe_ops.hip)
                return moe_mix(logits, outs)
        experts = torch.stack(outs, dim=1)  # [B, E, H]: the reference's literal form
        return [torch.matmul(torch.softmax(lg, dim=1).unsqueeze(1), experts).squeeze(1) for lg in logits]


class ConfigMMoE(RankModel):
    """`mmoe {...}` (tzrec/models/mmoe.py:36-96): the MMoE module over the (single) feature group and
    a task tower (MLP + logits layer) per task.  Predictions and losses carry the tower name as
    suffix, as the reference's multi-task models do."""

    def __init__(self, spec: PipelineSpec, device=None, sparse_optimizer=None) -> None:
        super().__init__(spec, device, sparse_optimizer)
        eg, m = self.embedding_group, spec.model
        self._group = eg.group_names()[0]
        d_in = eg.group_total_dim(self._group)
        expert = mlp_kwargs(m.one("expert_mlp"))
        hidden = expert["hidden_units"]
        gate = mlp_kwargs(m.one("gate_mlp")) if m.has("gate_mlp") else None
        self._towers = [(str(t.one("tower_name")), str(t.one("label_name"))) for t in m.many("task_towers")]
        self.mmoe = MMoE(d_in, expert, int(m.one("num_expert")), len(self._towers), gate)
        self.task_mlps = nn.ModuleList()
        self.task_outputs = nn.ModuleList()
        for t in m.many("task_towers"):
            if int(t.one("num_class", 1)) != 1:
                raise NotImplementedError("task towers with num_class > 1")
            d = hidden[-1]
            if t.has("mlp"):
                self.task_mlps.append(mlp_from_msg(d, t.one("mlp")))
                d = self.task_mlps[-1].output_dim()
            else:
                self.task_mlps.append(nn.Identity())
            self.task_outputs.append(OutputLinear(d, 1))
        if device is not None:
            for mod in (self.mmoe, self.task_mlps, self.task_outputs):
                mod.to(device)

    def forward(self, batch: Batch) -> Dict[str, torch.Tensor]:
        task_inputs = self.mmoe(self.build_input(batch)[self._group])
        out: Dict[str, torch.Tensor] = {}
        for i, (tower, _) in enumerate(self._towers):
            logits = self.task_outputs[i](self.task_mlps[i](task_inputs[i])).squeeze(1)
            out[f"logits_{tower}"], out[f"probs_{tower}"] = logits, torch.sigmoid(logits)
        return out

    def loss(self, predictions: Dict[str, torch.Tensor], batch: Batch) -> Dict[str, torch.Tensor]:
        from .dlrm import bce_with_logits

        return {f"binary_cross_entropy_{tower}": bce_with_logits(predictions[f"logits_{tower}"], batch.labels[label])
                for tower, label in self._towers}


_MODELS = {"dlrm": ConfigDLRM, "deepfm": ConfigDeepFM, "multi_tower_din": ConfigMultiTowerDIN, "mmoe": ConfigMMoE}


def build_rank_model(spec: PipelineSpec, device=None, sparse_optimizer=None, process_group=None,
                     plan: Optional[Dict[str, dict]] = None, use_planner: bool = False, exchange: str = "exact") -> RankModel:
    """Class chosen by the model_config oneof name, as tzrec/main.py:151-153 does.  With
    `process_group` the embedding tables are sharded over its ranks (`plan`: planner.plan_tables output
    or None: the planner under the config's `embedding_constraints` / `global_embedding_constraints` when it has
    any or when `use_planner` is set, else the size heuristic) and the dense parameters are kept identical across
    ranks."""
    if spec.model_name not in _MODELS:
        raise NotImplementedError(f"model {spec.model_name!r} is outside SURVEY.md section 8")
    RankModel._build_pg, RankModel._build_plan, RankModel._build_use_planner = process_group, plan, use_planner
    RankModel._build_exchange = exchange  # "capacity": fixed-slice ids exchange for the sharded pooled tables (sharding.py)
    try:
        model = _MODELS[spec.model_name](spec, device, sparse_optimizer)
    finally:
        RankModel._build_pg, RankModel._build_plan, RankModel._build_use_planner = None, None, False
        RankModel._build_exchange = "exact"
    model.sync_dense_parameters()
    return model
