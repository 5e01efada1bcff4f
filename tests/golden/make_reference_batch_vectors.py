"""Generate tests/golden/reference_batch_vectors.json by RUNNING the reference's batch assembly.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_reference_batch_vectors.py

SURVEY section 8 rows a2 / a3: `DataParser.parse` + `DataParser.to_batch`
(/root/reference/tzrec/datasets/data_parser.py:60-166,200-344,400-594) turn parsed columns into the
flat `<key>.values / .lengths / .weights / .key_lengths` tensor dict and then into the `Batch`
(KeyedJaggedTensor per data group, `sequence_mulval_lengths`, `sequence_dense_features`,
KeyedTensor, labels, sample weights).  The real class is driven here; what it needs around it:

* feature objects: the reference's feature classes need the protoc output, so a stand-in exposes the
  handful of attributes DataParser reads (name, is_sparse, is_sequence, value_dim, is_weighted,
  data_group, inputs, fg_mode = FG_NONE, ...) and a `parse` that calls the reference's own
  `_parse_fg_encoded_*_impl` functions, exactly what BaseFeature._parse does for FG_NONE
  (tzrec/features/feature.py:897-931);
* torchrec's KeyedJaggedTensor / KeyedTensor / JaggedTensor are containers here: recording
  stand-ins keep the constructor arguments the reference passes (that IS the output under test);
* every other absent dependency gets the attribute-less placeholder of
  make_reference_module_vectors.py.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import pyarrow as pa
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_reference_module_vectors as mk  # noqa: E402
import make_reference_parser_vectors as mp  # noqa: E402


class _Rec:
    def __init__(self, *args, **kw):
        assert not args, "the reference passes keywords"
        self.kw = kw


class KeyedJaggedTensor(_Rec):
    pass


class KeyedTensor(_Rec):
    pass


class JaggedTensor(_Rec):
    pass


class Pipelineable:
    pass


def install():
    mk.install_reference_imports()
    sys.meta_path.insert(0, mp._MoreAbsentDeps())
    for name in ("torchrec", "torchrec.sparse", "torchrec.sparse.jagged_tensor", "torchrec.streamable"):
        m = mk._Placeholder(name)
        m.__path__ = []
        m.KeyedJaggedTensor, m.KeyedTensor, m.JaggedTensor, m.Pipelineable = KeyedJaggedTensor, KeyedTensor, JaggedTensor, Pipelineable
        sys.modules[name] = m


def _t(x):
    if x is None:
        return None
    a = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    return {"dtype": str(a.dtype), "shape": list(a.shape), "data": a.reshape(-1).tolist()}


def main():
    install()
    F = importlib.import_module("tzrec.features.feature")
    DP = importlib.import_module("tzrec.datasets.data_parser")
    U = importlib.import_module("tzrec.datasets.utils")
    Mode = importlib.import_module("tzrec.constant").Mode
    S = chr(3)

    class Feature:
        """what DataParser reads from a BaseFeature, FG_NONE"""

        def __init__(self, name, sparse, sequence=False, value_dim=None, weighted=False, default=None, seq_delim=";"):
            self.name, self.is_sparse, self.is_sequence, self.is_weighted = name, sparse, sequence, weighted
            self.value_dim = value_dim if value_dim is not None else (1 if (sequence or not sparse) else 0)
            self.default, self.sequence_delim = default, seq_delim
            self.stub_type, self.is_user_feat, self.is_neg = False, False, False
            self.data_group, self.inputs, self.fg_mode = U.BASE_DATA_GROUP, [name], F.FgMode.FG_NONE

        def parse(self, input_data, is_training=False):
            feat = input_data[self.name]
            kw = {"default_value": self.default}
            if self.is_sequence:
                if self.is_sparse:
                    return F._parse_fg_encoded_sequence_sparse_feature_impl(self.name, feat, sequence_delim=self.sequence_delim, **kw)
                return F._parse_fg_encoded_sequence_dense_feature_impl(self.name, feat, sequence_delim=self.sequence_delim,
                                                                       value_dim=self.value_dim, **kw)
            if self.is_sparse:
                return F._parse_fg_encoded_sparse_feature_impl(self.name, feat, is_weighted=self.is_weighted, **kw)
            return F._parse_fg_encoded_dense_feature_impl(self.name, feat, **kw)

    cases = []

    def run(tag, feats, columns, labels=(), sample_weights=()):
        dp = DP.DataParser(feats, labels=list(labels), sample_weights=list(sample_weights), mode=Mode.TRAIN)
        data = {k: mp.to_arrow(rows, t) for k, (t, rows) in columns.items()}
        flat = dp.parse(data)
        b = dp.to_batch(flat)
        dg = U.BASE_DATA_GROUP
        rec = {"tag": tag,
               "features": [{"name": f.name, "sparse": f.is_sparse, "sequence": f.is_sequence, "value_dim": f.value_dim,
                             "weighted": f.is_weighted, "default": f.default} for f in feats],
               "columns": {k: {"type": t, "rows": rows} for k, (t, rows) in columns.items()},
               "labels": list(labels), "sample_weights": list(sample_weights),
               "flat": {k: _t(v) for k, v in flat.items()}}
        kjt = b.sparse_features[dg].kw
        rec["kjt"] = {"keys": list(kjt["keys"]), "values": _t(kjt["values"]), "lengths": _t(kjt["lengths"]),
                      "weights": _t(kjt.get("weights")), "stride": int(kjt["stride"]), "length_per_key": list(kjt["length_per_key"])}
        if dg in b.sequence_mulval_lengths:
            m = b.sequence_mulval_lengths[dg].kw
            rec["mulval"] = {"keys": list(m["keys"]), "values": _t(m["values"]), "lengths": _t(m["lengths"])}
        if dg in b.dense_features:
            d = b.dense_features[dg].kw
            rec["dense"] = {"keys": list(d["keys"]), "length_per_key": list(d["length_per_key"]), "values": _t(d["values"])}
        rec["seq_dense"] = {k: {"values": _t(v.kw["values"]), "lengths": _t(v.kw["lengths"])} for k, v in b.sequence_dense_features.items()}
        rec["batch_labels"] = {k: _t(v) for k, v in b.labels.items()}
        rec["batch_sample_weights"] = {k: _t(v) for k, v in b.sample_weights.items()}
        cases.append(rec)

    rng = np.random.default_rng(7)
    B = 6
    run("criteo_like",
        [Feature("c0", True), Feature("c1", True), Feature("c2", True), Feature("i0", False), Feature("i13", False, value_dim=3)],
        {"c0": ("int64", [int(x) for x in rng.integers(0, 50, B)]), "c1": ("int64", [3, None, 5, 7, None, 1]),
         "c2": ("string", ["4", "", "9", "1", "2", "3"]),
         "i0": ("float32", [float(x) for x in rng.integers(0, 9, B) / 4]),
         "i13": ("string", [S.join(str(float(v)) for v in rng.integers(0, 9, 3) / 4) for _ in range(B)]),
         "label": ("int64", [0, 1, 0, 0, 1, 0])},
        labels=["label"])
    run("weighted_group",
        [Feature("tags", True, weighted=True, default=[0]), Feature("cats", True), Feature("uid", True, default=[7])],
        {"tags": ("string", [f"3:0.5{S}4:1.5", "", None, "9:2.0", f"1:0.25{S}2:0.5{S}3:0.75", "5:1.0"]),
         "cats": ("string", [f"1{S}2", "3", "", None, f"4{S}5{S}6", "7"]),
         "uid": ("int64", [10, None, 12, 13, None, 15]),
         "clk": ("float32", [0.0, 1.0, 0.0, 1.0, 1.0, 0.0]), "w": ("float32", [1.0, 0.5, 2.0, 1.0, 1.0, 0.25])},
        labels=["clk"], sample_weights=["w"])
    run("sequences",
        [Feature("item", True), Feature("hist__cat", True, sequence=True), Feature("hist__tags", True, sequence=True, value_dim=0),
         Feature("hist__dwell", False, sequence=True, value_dim=2), Feature("price", False)],
        {"item": ("int64", [3, 5, 3, 9]),
         "hist__cat": ("string", ["1;2;3", "4", "1;1;2;3;5", "0;6"]),
         "hist__tags": ("string", [f"1{S}2;3;4{S}5{S}6", "7", f"1;1{S}1;2;3{S}12;5", f"0;9{S}10"]),
         "hist__dwell": ("list<list<float32>>", [[[float(x) for x in rng.integers(-4, 5, 2) / 4] for _ in range(n)] for n in (3, 1, 5, 2)]),
         "price": ("float64", [0.5, 1.25, 2.0, 0.0]),
         "buy": ("int32", [1, 0, 0, 1])},
        labels=["buy"])
    run("empty_sequences_and_defaults",
        [Feature("q", True), Feature("s__a", True, sequence=True, default=[0]), Feature("s__b", True, sequence=True, value_dim=0)],
        {"q": ("int64", [1, 2, 3]),
         "s__a": ("string", ["", "4;5", None]),
         "s__b": ("list<list<int64>>", [[[1, 2], [3]], [], [[4], [5, 6, 7], [8]]])})

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_batch_vectors.json")
    json.dump({"generator": "tests/golden/make_reference_batch_vectors.py", "cases": cases}, open(path, "w"))
    print(f"wrote {path}: {len(cases)} batches, {os.path.getsize(path)} bytes")
    for c in cases:
        print("  ", c["tag"], "kjt keys", c["kjt"]["keys"], "flat keys", len(c["flat"]))


if __name__ == "__main__":
    main()
