"""Generate tests/golden/reference_parser_vectors.json by RUNNING the reference's column parsers.

Run in the authoring container only (needs /root/reference):

    python tests/golden/make_reference_parser_vectors.py

The four functions of SURVEY section 8 row a1 are plain pyarrow / numpy code:

    tzrec/features/feature.py:80-166    _parse_fg_encoded_sparse_feature_impl
    tzrec/features/feature.py:169-214   _parse_fg_encoded_dense_feature_impl
    tzrec/features/feature.py:217-278   _parse_fg_encoded_sequence_sparse_feature_impl
    tzrec/features/feature.py:281-343   _parse_fg_encoded_sequence_dense_feature_impl

Their module imports torchrec, pyfg and friends at the top; those (and the one tzrec helper that
executes torchrec calls at import time, tzrec/utils/dynamicemb_util.py) get the same attribute-less
placeholders make_reference_module_vectors.py uses -- none is touched by the four functions.

Every case stores the column (python rows + an arrow type tag), the arguments, and what the
reference returned (or that it raised).  Random columns cover: null rows, empty strings / lists,
multi-value rows, weighted `id:w` tokens, maps, int columns, with and without a default.
"""
import importlib
import importlib.abc
import importlib.machinery
import json
import os
import sys

import numpy as np
import pyarrow as pa

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_reference_module_vectors as mk  # noqa: E402


class _MoreAbsentDeps(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    TOP = ("pyfg", "fbgemm_gpu", "graphlearn", "odps", "common_io", "dynamicemb", "alibabacloud_credentials", "tensordict")
    EXACT = ("tzrec.utils.dynamicemb_util",)

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.TOP or name in self.EXACT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return mk._Placeholder(spec.name)

    def exec_module(self, module):
        pass


ARROW_TYPES = {
    "string": pa.string(),
    "int32": pa.int32(),
    "int64": pa.int64(),
    "float32": pa.float32(),
    "float64": pa.float64(),
    "list<int64>": pa.list_(pa.int64()),
    "list<int32>": pa.list_(pa.int32()),
    "list<string>": pa.list_(pa.string()),
    "list<float32>": pa.list_(pa.float32()),
    "list<list<int64>>": pa.list_(pa.list_(pa.int64())),
    "list<list<float32>>": pa.list_(pa.list_(pa.float32())),
    "map<int64,float32>": pa.map_(pa.int64(), pa.float32()),
    "map<string,float32>": pa.map_(pa.string(), pa.float32()),
}


def to_arrow(rows, tag):
    if tag.startswith("map"):
        rows = [None if r is None else [tuple(kv) for kv in r] for r in rows]
    return pa.array(rows, type=ARROW_TYPES[tag])


def _js(a):
    if a is None:
        return None
    a = np.asarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape), "data": a.reshape(-1).tolist()}


def main():
    mk.install_reference_imports()
    sys.meta_path.insert(0, _MoreAbsentDeps())
    F = importlib.import_module("tzrec.features.feature")
    rng = np.random.default_rng(20260925)
    SEP = chr(3)
    cases = []

    def ids(n):
        return [int(x) for x in rng.integers(0, 1000, size=n)]

    def run(kind, tag, rows, args):
        fn = {"sparse": F._parse_fg_encoded_sparse_feature_impl, "dense": F._parse_fg_encoded_dense_feature_impl,
              "seq_sparse": F._parse_fg_encoded_sequence_sparse_feature_impl,
              "seq_dense": F._parse_fg_encoded_sequence_dense_feature_impl}[kind]
        case = {"kind": kind, "type": tag, "rows": rows, "args": args}
        try:
            r = fn("f", to_arrow(rows, tag), **args)
        except Exception as e:  # recorded: the product must refuse these too
            case["raises"] = type(e).__name__
            cases.append(case)
            return
        out = {"values": _js(r.values)}
        for k in ("lengths", "weights", "seq_lengths"):
            if hasattr(r, k):
                out[k] = _js(getattr(r, k))
        case["out"] = out
        cases.append(case)

    B = 12

    def maybe(p_null, p_empty, empty, make):
        u = rng.random()
        return None if u < p_null else (empty if u < p_null + p_empty else make())

    for default in (None, [0], [7, 9]):
        for sep in (SEP, ","):
            # string ids, multi-valued
            rows = [maybe(0.15, 0.15, "", lambda: sep.join(map(str, ids(rng.integers(1, 4))))) for _ in range(B)]
            run("sparse", "string", rows, {"multival_sep": sep, "default_value": default})
            # weighted string tokens
            rows = [maybe(0.15, 0.15, "", lambda: sep.join(f"{i}:{rng.integers(1, 9) / 4}" for i in ids(rng.integers(1, 4)))) for _ in range(B)]
            run("sparse", "string", rows, {"multival_sep": sep, "default_value": default, "is_weighted": True})
        rows = [maybe(0.15, 0.15, [], lambda: ids(rng.integers(1, 4))) for _ in range(B)]
        run("sparse", "list<int64>", rows, {"default_value": default})
        run("sparse", "list<int32>", rows, {"default_value": default})
        rows = [maybe(0.15, 0.15, [], lambda: [str(i) for i in ids(rng.integers(1, 4))]) for _ in range(B)]
        run("sparse", "list<string>", rows, {"default_value": default})
        rows = [maybe(0.15, 0.15, [], lambda: [f"{i}:{rng.integers(1, 9) / 4}" for i in ids(rng.integers(1, 4))]) for _ in range(B)]
        run("sparse", "list<string>", rows, {"default_value": default, "is_weighted": True})
        rows = [maybe(0.2, 0.0, None, lambda: [[i, float(rng.integers(1, 9) / 4)] for i in sorted(set(ids(rng.integers(1, 4))))]) for _ in range(B)]
        run("sparse", "map<int64,float32>", rows, {"default_value": default})
        rows = [maybe(0.2, 0.0, None, lambda: [[str(i), float(rng.integers(1, 9) / 4)] for i in sorted(set(ids(rng.integers(1, 4))))]) for _ in range(B)]
        run("sparse", "map<string,float32>", rows, {"default_value": default})
        for tag in ("int32", "int64"):
            rows = [maybe(0.25, 0.0, None, lambda: ids(1)[0]) for _ in range(B)]
            run("sparse", tag, rows, {"default_value": default[:1] if default else None})
    # no-null fast paths
    run("sparse", "string", [SEP.join(map(str, ids(2))) for _ in range(B)], {"default_value": None})
    run("sparse", "int64", ids(B), {"default_value": None})
    run("sparse", "float32", [0.5] * 4, {"default_value": None})  # refused by the reference

    for default in (None, [0], [3, 4]):
        rows = [maybe(0.15, 0.15, "", lambda: ";".join(SEP.join(map(str, ids(rng.integers(1, 3)))) for _ in range(rng.integers(1, 5)))) for _ in range(B)]
        run("seq_sparse", "string", rows, {"default_value": default})
        rows = [maybe(0.15, 0.0, "", lambda: "|".join(",".join(map(str, ids(rng.integers(1, 3)))) for _ in range(rng.integers(1, 5)))) for _ in range(B)]
        run("seq_sparse", "string", rows, {"sequence_delim": "|", "multival_sep": ",", "default_value": default})
        rows = [maybe(0.15, 0.15, [], lambda: ids(rng.integers(1, 5))) for _ in range(B)]
        run("seq_sparse", "list<int64>", rows, {"default_value": default})
        rows = [maybe(0.15, 0.15, [], lambda: [ids(rng.integers(1, 3)) for _ in range(rng.integers(1, 5))]) for _ in range(B)]
        run("seq_sparse", "list<list<int64>>", rows, {"default_value": default})

    def fl(n):
        return [float(x) for x in (rng.integers(-8, 8, size=n) / 4)]

    for default in (None, [0.5, 0.25]):
        rows = [maybe(0.0 if default is None else 0.2, 0.0 if default is None else 0.15, "", lambda: SEP.join(map(str, fl(2)))) for _ in range(B)]
        run("dense", "string", rows, {"default_value": default})
        rows = [maybe(0.0 if default is None else 0.2, 0.0 if default is None else 0.15, [], lambda: fl(2)) for _ in range(B)]
        run("dense", "list<float32>", rows, {"default_value": default})
    for tag in ("float32", "float64", "int64"):
        for default in (None, [1.5]):
            rows = [maybe(0.0 if default is None else 0.25, 0.0, None, lambda: (ids(1)[0] if tag == "int64" else fl(1)[0])) for _ in range(B)]
            run("dense", tag, rows, {"default_value": default})
    for default in (None, [0.5, 0.25]):
        rows = [maybe(0.15 if default else 0.0, 0.1 if default else 0.0, "", lambda: ";".join(SEP.join(map(str, fl(2))) for _ in range(rng.integers(1, 4)))) for _ in range(B)]
        run("seq_dense", "string", rows, {"value_dim": 2, "default_value": default})
        rows = [maybe(0.15 if default else 0.0, 0.1 if default else 0.0, [], lambda: [fl(2) for _ in range(rng.integers(1, 4))]) for _ in range(B)]
        run("seq_dense", "list<list<float32>>", rows, {"value_dim": 2, "default_value": default})
    rows = [fl(rng.integers(1, 4)) for _ in range(B)]
    run("seq_dense", "list<float32>", rows, {"value_dim": 1, "default_value": None})

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_parser_vectors.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_reference_parser_vectors.py", "cases": cases}, f)
    n_raise = sum("raises" in c for c in cases)
    print(f"wrote {path}: {len(cases)} cases ({n_raise} where the reference raises), {os.path.getsize(path)} bytes")
    for c in cases:
        if "raises" in c:
            print("  raises:", c["kind"], c["type"], c["args"], c["raises"])


if __name__ == "__main__":
    main()
