"""Host index stage against OUTPUTS OF THE REFERENCE'S OWN PARSERS.

tests/golden/reference_parser_vectors.json was produced by running
/root/reference/tzrec/features/feature.py:80-343 (the four `_parse_fg_encoded_*_impl` functions) on
random arrow columns of every supported type (tests/golden/make_reference_parser_vectors.py, run in
the authoring container).  Bit-exact for ids / lengths, exact for float32 values.
"""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from torcheasyrec_amd import data_parser as dp  # noqa: E402

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_parser_vectors.json")))["cases"]


def _arrow(case):
    from make_reference_parser_vectors import to_arrow

    return to_arrow(case["rows"], case["type"])


def _arr(js):
    return None if js is None else np.asarray(js["data"], dtype=js["dtype"]).reshape(js["shape"])


def _id(case):
    a = case["args"]
    return f"{case['kind']}-{case['type']}-d{a.get('default_value')}-w{int(bool(a.get('is_weighted')))}-s{ord(a.get('multival_sep', chr(3))[0])}"


@pytest.mark.parametrize("case", _G, ids=[f"{i}-{_id(c)}" for i, c in enumerate(_G)])
def test_parser_matches_reference_output(case):
    fn = {"sparse": dp.parse_sparse_column, "dense": dp.parse_dense_column, "seq_sparse": dp.parse_sequence_column,
          "seq_dense": dp.parse_sequence_dense_column}[case["kind"]]
    col = _arrow(case)
    if "raises" in case:
        with pytest.raises(ValueError):
            fn("f", col, **case["args"])
        return
    got = fn("f", col, **case["args"])
    want = case["out"]
    v = _arr(want["values"])
    assert got.values.dtype == v.dtype, (got.values.dtype, v.dtype)
    np.testing.assert_array_equal(got.values, v)
    if "lengths" in want:
        np.testing.assert_array_equal(got.lengths, _arr(want["lengths"]))
    if "seq_lengths" in want:
        np.testing.assert_array_equal(got.seq_lengths, _arr(want["seq_lengths"]))
    if "weights" in want:
        w = _arr(want["weights"])
        if w is None:
            assert got.weights is None
            return
        assert got.weights.dtype == np.float32
        d = case["args"].get("default_value")
        if d is not None and len(d) > 1:
            # KNOWN DIVERGENCE.  The reference fills a missing row's ids with the whole default list but
            # its weights with `[1.0]` (feature.py:141-144), so for a multi-id default its weights array
            # is shorter than its values array and every later weight is misaligned.  This package
            # emits one 1.0 per default id; the expectation is the reference's output with that repair.
            missing = [r is None or r == "" or r == [] for r in case["rows"]]
            fixed, k = [], 0
            for miss, n in zip(missing, _arr(want["lengths"])):
                if miss:
                    fixed += [1.0] * int(n)
                    k += 1
                else:
                    fixed += list(w[k:k + int(n)])
                    k += int(n)
            assert k == len(w)
            w = np.asarray(fixed, np.float32)
        assert len(got.weights) == len(got.values)
        np.testing.assert_array_equal(got.weights, w)


_SPARSE = [c for c in _G if c["kind"] == "sparse" and "out" in c]


@pytest.mark.parametrize("case", _SPARSE, ids=[f"{i}-{_id(c)}" for i, c in enumerate(_SPARSE)])
def test_oracle_parse_matches_reference_output(case):
    """pins oracle.parse_sparse_feature (the restatement the KJT / bucketize tests feed from)"""
    from oracle import tzrec_oracle as orc

    rows, a = case["rows"], case["args"]
    if case["type"].startswith("map"):
        rows = [None if r is None else {k: v for k, v in r} for r in rows]
    d = a.get("default_value")
    v, l, w = orc.parse_sparse_feature(rows, d, a.get("multival_sep", chr(3)), bool(a.get("is_weighted")))
    np.testing.assert_array_equal(v, _arr(case["out"]["values"]))
    np.testing.assert_array_equal(l, _arr(case["out"]["lengths"]))
    wr = _arr(case["out"].get("weights"))
    assert (w is None) == (wr is None)
    if w is not None and not (d is not None and len(d) > 1):  # multi-id default: see the divergence note above
        np.testing.assert_array_equal(w, wr)
