#!/usr/bin/env python3
"""Golden vectors of the index stage, transcribed from the reference's own tests.

The reference cannot be imported here (torchrec / fbgemm_gpu / pyfg / generated protos are absent,
SURVEY.md section 0), so the vectors are the literals of its unit tests.  This script is what
produced tests/golden/reference_index_vectors.json; when /root/reference is present it also checks
that every literal it transcribes still appears verbatim in the cited reference test file, so the
fixture cannot drift from the reference silently.

    python tests/golden/make_reference_index_vectors.py          # rewrite the JSON
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_index_vectors.json")
S3 = "\x03"

CASES = {
    # tzrec/features/id_feature_test.py:41-69  test_fg_encoded_id_feature
    # [input column, fg_encoded_default_value, expected values, expected lengths]
    "id_feature_parse": {
        "source": "tzrec/features/id_feature_test.py:41-69",
        "cases": [
            {"input": ["1" + S3 + "2", "", None, "3"], "default": "", "values": [1, 2, 3], "lengths": [2, 0, 0, 1]},
            {"input": ["1" + S3 + "2", "", None, "3"], "default": "0", "values": [1, 2, 0, 0, 3], "lengths": [2, 1, 1, 1]},
            {"input": [1, 2, None, 3], "default": "", "values": [1, 2, 3], "lengths": [1, 1, 0, 1]},
            {"input": [1, 2, None, 3], "default": "0", "values": [1, 2, 0, 3], "lengths": [1, 1, 1, 1]},
        ],
        "verify": [
            '[["1\\x032", "", None, "3"], "", [1, 2, 3], [2, 0, 0, 1]]',
            '[["1\\x032", "", None, "3"], "0", [1, 2, 0, 0, 3], [2, 1, 1, 1]]',
            '[[1, 2, None, 3], "", [1, 2, 3], [1, 1, 0, 1]]',
            '[[1, 2, None, 3], "0", [1, 2, 0, 3], [1, 1, 1, 1]]',
        ],
    },
    # tzrec/features/id_feature_test.py:152-188  test_fg_encoded_with_weighted (map input)
    "id_feature_parse_weighted": {
        "source": "tzrec/features/id_feature_test.py:152-188",
        "cases": [
            {"input": [{"1": 1.0}, {"2": 1.5, "3": 2.0}, {"4": 2.5}], "values": [1, 2, 3, 4],
             "lengths": [1, 2, 1], "weights": [1.0, 1.5, 2.0, 2.5]},
        ],
        "verify": ['[{"1": 1.0}, {"2": 1.5, "3": 2.0}, {"4": 2.5}]', "expected_values = [1, 2, 3, 4]",
                   "expected_lengths = [1, 2, 1]", "expected_weights = [1.0, 1.5, 2.0, 2.5]"],
    },
    # tzrec/datasets/data_parser_test.py:36-154  test_nofg: parse + to_batch -> KeyedJaggedTensor
    "data_parser_nofg": {
        "source": "tzrec/datasets/data_parser_test.py:36-154",
        "file": "tzrec/datasets/data_parser_test.py",
        "columns": {
            "cat_a": {"input": [1, 2, 3], "default": None, "sep": S3},
            "tag_b": {"input": ["4" + S3 + "5", "", "6"], "default": None, "sep": S3},
            "click_seq__cat_a": {"input": ["10;11;12", "13", ""], "default": None, "sep": ";"},
        },
        "kjt": {"keys": ["cat_a", "tag_b", "click_seq__cat_a"],
                "values": [1, 2, 3, 4, 5, 6, 10, 11, 12, 13],
                "lengths": [1, 1, 1, 2, 0, 1, 3, 1, 0]},
        "verify": ["values=torch.tensor([1, 2, 3, 4, 5, 6, 10, 11, 12, 13])",
                   "lengths=torch.tensor([1, 1, 1, 2, 0, 1, 3, 1, 0], dtype=torch.int32)",
                   'keys=["cat_a", "tag_b", "click_seq__cat_a"]'],
    },
    # tzrec/datasets/data_parser_test.py:156-330  test_fg_encoded_id_with_weight
    "data_parser_weighted": {
        "source": "tzrec/datasets/data_parser_test.py:156-330",
        "file": "tzrec/datasets/data_parser_test.py",
        "columns": {
            "cat_a": {"input": [{"1": 2.0}, {"2": 1.0}, {"3": 3.5}], "default": None, "sep": S3, "weighted": True},
            "cat_a1": {"input": [{"1": 2.0}, None, {"3": 3.5}], "default": [0], "sep": S3, "weighted": True},
            "tag_b": {"input": ["4:2.3" + S3 + "5:2.4", "", "6:2.5"], "default": None, "sep": S3, "weighted": True},
            "tag_b1": {"input": ["4:2.3" + S3 + "5:2.4", "", "6:2.5"], "default": [0], "sep": S3, "weighted": True},
            "click_seq__cat_a": {"input": ["10;11;12", "13", ""], "default": None, "sep": ";"},
        },
        "kjt": {"keys": ["cat_a", "cat_a1", "tag_b", "tag_b1", "click_seq__cat_a"],
                "values": [1, 2, 3, 1, 0, 3, 4, 5, 6, 4, 5, 0, 6, 10, 11, 12, 13],
                "lengths": [1, 1, 1, 1, 1, 1, 2, 0, 1, 2, 1, 1, 3, 1, 0],
                "weights": [2.0, 1.0, 3.5, 2.0, 1.0, 3.5, 2.3, 2.4, 2.5, 2.3, 2.4, 1.0, 2.5, 1, 1, 1, 1]},
        "verify": ["[1, 2, 3, 1, 0, 3, 4, 5, 6, 4, 5, 0, 6, 10, 11, 12, 13]",
                   "[1, 1, 1, 1, 1, 1, 2, 0, 1, 2, 1, 1, 3, 1, 0], dtype=torch.int32",
                   "expected_tag_b1_values = torch.tensor([4, 5, 0, 6], dtype=torch.int64)",
                   "expected_cat_a1_values = torch.tensor([1, 0, 3], dtype=torch.int64)"],
    },
    # shape fixtures the reference pins for the floating-point stages (values are NOT pinned there)
    "shape_fixtures": {
        "source": "tzrec/modules/interaction_test.py:43-54, tzrec/modules/fm_test.py:23-32, "
                  "tzrec/modules/embedding_test.py:241-256",
        "interaction": {"feature_num": 4, "batch": 10, "output": [10, 6]},
        "fm": {"input": [4, 26, 16], "output": [4, 16]},
    },
}


def _norm(s: str) -> str:
    return re.sub(r"\s+", "", s)


def verify_against_reference() -> int:
    checked = 0
    for name, case in CASES.items():
        src = case["source"].split(":")[0].split(",")[0].strip()
        path = os.path.join(REF, src)
        if not os.path.exists(path):
            continue
        text = _norm(open(path).read())
        for lit in case.get("verify", []):
            assert _norm(lit) in text, f"{name}: literal not found in {src}: {lit}"
            checked += 1
    return checked


def main():
    n = verify_against_reference() if os.path.isdir(REF) else 0
    out = {k: {kk: vv for kk, vv in v.items() if kk != "verify"} for k, v in CASES.items()}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(f"wrote {OUT}; {n} literals verified against {REF}")


if __name__ == "__main__":
    main()
