"""Full-size pipeline configs of the BASELINE.json model families, as TEXT in the reference's config format.

`bench.py` times a train step of each (`secondary.deepfm_criteo_b8192`, `din_taobao_b8192`, `mmoe_zch_b8192`) and the
examples / tests can feed them to `config.load_pipeline_spec` like any `.config` file.  The texts are generated here from
the facts of the reference's example configs -- feature names, bucket counts, embedding dims, group membership, tower
widths -- not copied from them; I/O paths, FG expressions and metrics (which this package does not consume) are left out.

  deepfm_criteo()            examples/deepfm_criteo.config: 13 raw + 26 id features (the Criteo bucket counts of
                             dlrm_criteo.config), groups wide / fm / deep, deep {512, 256, 128}, final {64}
                             (/root/reference/tzrec/models/deepfm.py:72-108)
  multi_tower_din_taobao()   examples/multi_tower_din_taobao.config: 16 features in a DEEP group, a SEQUENCE group of three
                             item features with a 100-step click sequence, DIN attention MLP {256, 64}
                             (/root/reference/tzrec/modules/sequence.py:65-128)
  mmoe_taobao_zch()          examples/mmoe_taobao.config with the user id behind a zero-collision hash (`zch { zch_size
                             ... lfu {} }`, /root/reference/tzrec/protos/feature.proto:31-47): BASELINE.json configs[4],
                             "LFU eviction under 200M-row table"
"""
from __future__ import annotations

from typing import List, Sequence

from .criteo import CRITEO_ROWS, NUM_DENSE

_TRAIN = """train_config {
    sparse_optimizer { adagrad_optimizer { lr: 0.001 } constant_learning_rate { } }
    dense_optimizer { adam_optimizer { lr: 0.001 } constant_learning_rate { } }
    num_epochs: 1
}
"""

# (feature, num_buckets) of the Taobao display-advertising examples (multi_tower_din_taobao.config / mmoe_taobao.config)
TAOBAO_ID_FEATURES = [("user_id", 1141730), ("cms_segid", 98), ("cms_group_id", 14), ("final_gender_code", 3), ("age_level", 8),
                      ("pvalue_level", 5), ("shopping_level", 5), ("occupation", 3), ("new_user_class_level", 6),
                      ("adgroup_id", 846812), ("cate_id", 12961), ("campaign_id", 423438), ("customer", 255877), ("brand", 461498)]
TAOBAO_PRICE_BOUNDARIES = 98  # `price`: a raw feature bucketized by 98 boundaries -> 99 buckets, embedded
TAOBAO_PID_BUCKETS = 20       # `pid`: hash_bucket_size 20


def _data(batch_size: int, labels: Sequence[str]) -> str:
    lab = "".join(f'    label_fields: "{l}"\n' for l in labels)
    return f"data_config {{\n    batch_size: {batch_size}\n    dataset_type: ParquetDataset\n    fg_mode: FG_NONE\n{lab}    num_workers: 8\n}}\n"


def _group(name: str, feats: Sequence[str], kind: str) -> str:
    return ("    feature_groups {\n        group_name: \"%s\"\n" % name + "".join(f'        feature_names: "{f}"\n' for f in feats)
            + f"        group_type: {kind}\n    }}\n")


def deepfm_criteo(batch_size: int = 8192, rows: Sequence[int] = CRITEO_ROWS) -> str:
    ints = [f"int_{i}" for i in range(NUM_DENSE)]
    cats = [f"cat_{i}" for i in range(len(rows))]
    feats = "".join(f'feature_configs {{ raw_feature {{ feature_name: "{n}" }} }}\n' for n in ints)
    feats += "".join(f'feature_configs {{ id_feature {{ feature_name: "{n}" num_buckets: {r} embedding_dim: 16 }} }}\n' for n, r in zip(cats, rows))
    model = ("model_config {\n" + _group("wide", cats, "WIDE") + _group("fm", cats, "DEEP") + _group("deep", ints + cats, "DEEP")
             + "    deepfm {\n        deep { hidden_units: [512, 256, 128] }\n        final { hidden_units: [64] }\n    }\n"
             + "    losses { binary_cross_entropy {} }\n}\n")
    return _TRAIN + _data(batch_size, ["label"]) + feats + model


def _taobao_features(zch_user_rows: int = 0, user_rows: int = 0) -> List[str]:
    out = []
    for n, r in TAOBAO_ID_FEATURES:
        if n == "user_id" and zch_user_rows:
            out.append(f'feature_configs {{ id_feature {{ feature_name: "user_id" embedding_dim: 16 zch {{ zch_size: {zch_user_rows} '
                       f'eviction_interval: 1000 lfu {{}} }} }} }}\n')
        else:
            out.append(f'feature_configs {{ id_feature {{ feature_name: "{n}" num_buckets: {user_rows if (n == "user_id" and user_rows) else r} embedding_dim: 16 }} }}\n')
    bounds = ", ".join(f"{1.0 + 1.5 * i:.1f}" for i in range(TAOBAO_PRICE_BOUNDARIES))
    out.append(f'feature_configs {{ raw_feature {{ feature_name: "price" boundaries: [{bounds}] embedding_dim: 16 }} }}\n')
    out.append(f'feature_configs {{ id_feature {{ feature_name: "pid" hash_bucket_size: {TAOBAO_PID_BUCKETS} embedding_dim: 16 }} }}\n')
    return out


def multi_tower_din_taobao(batch_size: int = 8192, sequence_length: int = 100) -> str:
    deep = [n for n, _ in TAOBAO_ID_FEATURES] + ["price", "pid"]
    rows = dict(TAOBAO_ID_FEATURES)
    seq = "".join(f'        features {{ id_feature {{ feature_name: "{n}" num_buckets: {rows[n]} embedding_dim: 16 }} }}\n'
                  for n in ("adgroup_id", "cate_id", "brand"))
    feats = "".join(_taobao_features()) + ("feature_configs {\n    sequence_feature {\n        sequence_name: \"click_50_seq\"\n"
                                           f"        sequence_length: {sequence_length}\n        sequence_delim: \"|\"\n{seq}    }}\n}}\n")
    model = ("model_config {\n" + _group("deep", deep, "DEEP")
             + _group("seq", ["adgroup_id", "cate_id", "brand", "click_50_seq__adgroup_id", "click_50_seq__cate_id", "click_50_seq__brand"], "SEQUENCE")
             + "    multi_tower_din {\n        towers { input: 'deep' mlp { hidden_units: [512, 256, 128] } }\n"
               "        din_towers { input: 'seq' attn_mlp { hidden_units: [256, 64] } }\n        final { hidden_units: [64] }\n    }\n"
               "    losses { binary_cross_entropy {} }\n}\n")
    return _TRAIN + _data(batch_size, ["clk"]) + feats + model


def mmoe_taobao_zch(batch_size: int = 8192, zch_size: int = 200_000_000) -> str:
    names = ["user_id", "cms_segid", "cms_group_id", "final_gender_code", "age_level", "pvalue_level", "shopping_level", "occupation",
             "new_user_class_level", "pid", "adgroup_id", "cate_id", "campaign_id", "customer", "brand", "price"]
    model = ("model_config {\n" + _group("all", names, "DEEP")
             + "    mmoe {\n        expert_mlp { hidden_units: [512, 256, 128] }\n        num_expert: 3\n"
               "        task_towers { tower_name: \"ctr\" label_name: \"clk\" mlp { hidden_units: [256, 128, 64] } losses { binary_cross_entropy {} } }\n"
               "        task_towers { tower_name: \"cvr\" label_name: \"buy\" mlp { hidden_units: [256, 128, 64] } losses { binary_cross_entropy {} } }\n"
               "    }\n}\n")
    return _TRAIN + _data(batch_size, ["clk", "buy"]) + "".join(_taobao_features(zch_user_rows=zch_size)) + model
