"""Read tzrec pipeline configs (protobuf text format) without protoc / generated *_pb2 modules.

The reference loads ``examples/*.config`` with ``text_format.Merge`` into ``EasyRecConfig``
(/root/reference/tzrec/utils/config_util.py:25-48); the generated ``tzrec/protos/*_pb2.py`` cannot
be produced here (no protoc).  The hot path only needs the few fields listed
in SURVEY.md section 8 (feature_configs, model_config.feature_groups, the dlrm/deepfm blocks,
train_config.sparse_optimizer, data_config.batch_size), so this module parses the text format
generically and extracts those.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Any, List, Optional

from .embedding import SparseOptimizerConfig

_TOKEN = re.compile(r'\s*(?:(#[^\n]*)|("(?:\\.|[^"\\])*")|([{}\[\]:,;<>])|([^\s{}\[\]:,;<>"]+))')


class Msg(dict):
    """A parsed message: field name -> list of values (scalars or Msg), in file order."""

    def one(self, key: str, default: Any = None) -> Any:
        v = self.get(key)
        return v[-1] if v else default

    def many(self, key: str) -> List[Any]:
        return self.get(key, [])

    def has(self, key: str) -> bool:
        return key in self


def _scalar(tok: str) -> Any:
    if tok.startswith('"') or tok.startswith("'"):  # text format allows either quote
        return bytes(tok[1:-1], "utf-8").decode("unicode_escape")
    if tok in ("true", "True"):
        return True
    if tok in ("false", "False"):
        return False
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok  # enum identifier


def parse_text_proto(text: str) -> Msg:
    toks: List[str] = []
    pos = 0
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"cannot tokenize config at offset {pos}: {text[pos:pos + 30]!r}")
        pos = m.end()
        if m.group(1):
            continue
        toks.append(m.group(2) or m.group(3) or m.group(4))
    i = 0

    def parse_msg(end: Optional[str]) -> Msg:
        nonlocal i
        out = Msg()
        while i < len(toks):
            t = toks[i]
            if t == end:
                i += 1
                return out
            if t in (",", ";"):
                i += 1
                continue
            name = t
            i += 1
            if i < len(toks) and toks[i] == ":":
                i += 1
            t = toks[i]
            if t in ("{", "<"):
                i += 1
                out.setdefault(name, []).append(parse_msg("}" if t == "{" else ">"))
            elif t == "[":
                i += 1
                while toks[i] != "]":
                    if toks[i] == ",":
                        i += 1
                        continue
                    if toks[i] == "{":
                        i += 1
                        out.setdefault(name, []).append(parse_msg("}"))
                    else:
                        out.setdefault(name, []).append(_scalar(toks[i]))
                        i += 1
                i += 1
            else:
                out.setdefault(name, []).append(_scalar(t))
                i += 1
        if end is not None:
            raise ValueError("unbalanced braces in config")
        return out

    return parse_msg(None)


@dataclass
class FeatureSpec:
    """What the hot path needs from a tzrec feature (tzrec/features/feature.py:586-662,
    id_feature.py:52-87, raw_feature.py:35-48)."""

    name: str
    kind: str  # "id_feature" | "raw_feature" | ...
    is_sparse: bool
    embedding_dim: int = 0
    num_embeddings: int = 0
    embedding_name: Optional[str] = None
    pooling: str = "sum"
    value_dim: int = 1
    trainable: bool = True
    is_sequence: bool = False  # sub-feature of a `sequence_feature` block, named <sequence_name>__<feature_name>
    sequence_length: int = 0
    data_type: str = "FP32"  # feature config `data_type` (FP32 | FP16)
    zch: Optional[Msg] = None  # the raw `zch {...}` block (see zch.zch_config_from_msg)
    # `embedding_constraints { sharding_types: ... }` (feature.proto:6-13, features/feature.py:832-845)
    sharding_types: List[str] = field(default_factory=list)


@dataclass
class FeatureGroupSpec:
    group_name: str
    feature_names: List[str]
    group_type: str = "DEEP"  # DEEP | WIDE | SEQUENCE
    embedding_name_suffix: Optional[str] = None


@dataclass
class PipelineSpec:
    features: List[FeatureSpec] = field(default_factory=list)
    feature_groups: List[FeatureGroupSpec] = field(default_factory=list)
    model_name: str = ""
    model: Msg = field(default_factory=Msg)
    wide_embedding_dim: int = 0
    num_class: int = 1
    batch_size: int = 0
    sparse_optimizer: Optional[SparseOptimizerConfig] = None
    dense_lr: float = 1e-3
    label_fields: List[str] = field(default_factory=list)
    # the raw optimizer blocks (learning-rate schedules: lr_scheduler.create_scheduler)
    sparse_optimizer_block: Optional[Msg] = None
    dense_optimizer_block: Optional[Msg] = None
    # train_config.delta_embedding_dump_config (train.proto:86-111), as parsed (Msg)
    delta_embedding_dump_config: Optional[object] = None
    # train_config.global_embedding_constraints.sharding_types (train.proto:144, plan_util.py:170-179)
    global_sharding_types: List[str] = field(default_factory=list)
    gradient_accumulation_steps: int = 0  # train.proto:151
    grad_clipping: Optional[object] = None  # train.proto:153 -> optimizer.GradClippingConfig


def _num_embeddings(f: Msg, name: str) -> int:
    """IdFeature.num_embeddings precedence (tzrec/features/id_feature.py:64-87)."""
    if f.has("zch"):
        return int(f.one("zch").one("zch_size"))
    if f.has("dynamicemb"):
        return int(f.one("dynamicemb").one("max_capacity"))
    if f.has("hash_bucket_size"):
        return int(f.one("hash_bucket_size"))
    if f.has("num_buckets"):
        return int(f.one("num_buckets"))
    if f.has("vocab_list"):
        return len(f.many("vocab_list"))
    raise ValueError(f"IdFeature[{name}] must set hash_bucket_size or num_buckets or vocab_list or zch.zch_size")


def sparse_optimizer_from_config(opt: Msg) -> SparseOptimizerConfig:
    """create_sparse_optimizer mapping (tzrec/optim/optimizer_builder.py:30-97) for the kinds this
    library fuses; field defaults from protos/optimizer.proto:76-139."""
    table = {"sgd_optimizer": "sgd", "adagrad_optimizer": "adagrad", "rowwise_adagrad_optimizer": "rowwise_adagrad",
             "adam_optimizer": "adam"}
    for key, kind in table.items():
        if opt.has(key):
            m = opt.one(key)
            return SparseOptimizerConfig(
                kind=kind, lr=float(m.one("lr", 0.002)), weight_decay=float(m.one("weight_decay", 0.0)),
                weight_decay_mode=str(m.one("weight_decay_mode", "NONE")).lower(),
                gradient_clipping=bool(m.one("gradient_clipping", False)),
                max_gradient=float(m.one("max_gradient", 1.0)),
                initial_accumulator_value=float(m.one("initial_accumulator_value", 0.0)),
                beta1=float(m.one("beta1", 0.9)), beta2=float(m.one("beta2", 0.999)),
            )
    raise ValueError(f"Unknown optimizer: {[k for k in opt.keys()]}")


def load_pipeline_spec(text: str) -> PipelineSpec:
    cfg = parse_text_proto(text)
    spec = PipelineSpec()
    def one_feature(kind: str, f: Msg, prefix: str = "", seq_len: int = 0) -> FeatureSpec:
        name = prefix + (f.one("feature_name") or f.one("sequence_name"))
        seq = dict(is_sequence=bool(prefix), sequence_length=seq_len)
        if kind == "id_feature":
            return FeatureSpec(
                name=name, kind=kind, is_sparse=True, embedding_dim=int(f.one("embedding_dim", 0)),
                num_embeddings=_num_embeddings(f, name), embedding_name=f.one("embedding_name"),
                pooling=str(f.one("pooling", "sum")).lower(), trainable=bool(f.one("trainable", True)),
                data_type=str(f.one("data_type", "FP32")).upper(),
                # id_feature.value_dim (tzrec/features/id_feature.py:42-50): configured, else 1 for a
                # sequence sub-feature (single id per step) and 0 = "any number of ids" otherwise
                value_dim=int(f.one("value_dim", 1 if prefix else 0)),
                zch=f.one("zch") if f.has("zch") else None, **seq)
        if kind == "raw_feature":
            nb = len(f.many("boundaries"))
            if nb:  # bucketized raw feature = a sparse id in [0, len(boundaries)]  (raw_feature.py:50-60)
                return FeatureSpec(name=name, kind=kind, is_sparse=True, embedding_dim=int(f.one("embedding_dim", 0)),
                                   num_embeddings=nb + 1, embedding_name=f.one("embedding_name"), **seq)
            return FeatureSpec(name=name, kind=kind, is_sparse=False, value_dim=int(f.one("value_dim", 1)), **seq)
        return FeatureSpec(name=name, kind=kind, is_sparse=f.has("embedding_dim"), embedding_dim=int(f.one("embedding_dim", 0)), **seq)

    _one_feature = one_feature

    def one_feature(kind: str, f: Msg, prefix: str = "", seq_len: int = 0) -> FeatureSpec:  # noqa: F811
        fs = _one_feature(kind, f, prefix, seq_len)
        if f.has("embedding_constraints"):
            fs.sharding_types = [str(t) for t in f.one("embedding_constraints").many("sharding_types")]
        return fs

    for fc in cfg.many("feature_configs"):
        (kind, body), = fc.items()
        f = body[-1]
        if kind == "sequence_feature":  # sub-features are named <sequence_name>__<feature_name> (feature.py)
            sname, slen = f.one("sequence_name"), int(f.one("sequence_length", 0))
            for sub in f.many("features"):
                (skind, sbody), = sub.items()
                spec.features.append(one_feature(skind, sbody[-1], prefix=f"{sname}__", seq_len=slen))
        else:
            spec.features.append(one_feature(kind, f))
    mc = cfg.one("model_config", Msg())
    for g in mc.many("feature_groups"):
        spec.feature_groups.append(FeatureGroupSpec(
            group_name=g.one("group_name"), feature_names=list(g.many("feature_names")),
            group_type=str(g.one("group_type", "DEEP")), embedding_name_suffix=g.one("embedding_name_suffix")))
    skip = {"feature_groups", "metrics", "losses", "num_class", "train_metrics", "variational_dropout",
            "kd", "use_pareto_loss_weight", "pareto"}
    for k, v in mc.items():
        if k not in skip and isinstance(v[-1], Msg):
            spec.model_name, spec.model = k, v[-1]
    spec.num_class = int(mc.one("num_class", 1))
    spec.wide_embedding_dim = int(spec.model.one("wide_embedding_dim", 0)) if spec.model else 0
    tc = cfg.one("train_config", Msg())
    if tc.has("sparse_optimizer"):
        spec.sparse_optimizer = sparse_optimizer_from_config(tc.one("sparse_optimizer"))
        spec.sparse_optimizer_block = tc.one("sparse_optimizer")
    if tc.has("dense_optimizer"):
        spec.dense_optimizer_block = tc.one("dense_optimizer")
        for _, v in tc.one("dense_optimizer").items():
            if isinstance(v[-1], Msg) and v[-1].has("lr"):
                spec.dense_lr = float(v[-1].one("lr"))
    spec.gradient_accumulation_steps = int(tc.one("gradient_accumulation_steps", 0))
    if tc.has("grad_clipping"):
        from .optimizer import grad_clipping_from_msg

        spec.grad_clipping = grad_clipping_from_msg(tc.one("grad_clipping"))
    if tc.has("global_embedding_constraints"):
        spec.global_sharding_types = [str(t) for t in tc.one("global_embedding_constraints").many("sharding_types")]
    if tc.has("delta_embedding_dump_config"):  # enable_delta_embedding_dump, tzrec/main.py:691
        # the block stays tzrec's to interpret (its DeltaEmbeddingDumper owns cadence / files); its presence
        # is what asks the model for a ModelDeltaTracker (delta_embedding_dump.py)
        spec.delta_embedding_dump_config = tc.one("delta_embedding_dump_config")
    dc = cfg.one("data_config", Msg())
    spec.batch_size = int(dc.one("batch_size", 0))
    spec.label_fields = list(dc.many("label_fields"))
    return spec
