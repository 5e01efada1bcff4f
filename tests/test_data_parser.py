"""Host-side index stage (torcheasyrec_amd/data_parser.py) against (1) the reference's own test
literals (tests/golden/reference_index_vectors.json) and (2) the oracle on random ragged columns of
every supported arrow type.  Bit-exact: ids, lengths and weights are integers / parsed literals."""
import json
import os
import sys

import numpy as np
import pyarrow as pa
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd.data_parser import (DataParser, parse_dense_column, parse_sequence_column,  # noqa: E402
                                          parse_sparse_column)

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_index_vectors.json")))
S3 = "\x03"


def _default(d):
    if d in (None, ""):
        return None
    return [int(x) for x in d] if isinstance(d, list) else [int(d)]


@pytest.mark.parametrize("case", G["id_feature_parse"]["cases"])
def test_id_feature_parse_golden(case):
    c = parse_sparse_column("f", case["input"], default_value=_default(case["default"]))
    assert c.values.tolist() == case["values"] and c.lengths.tolist() == case["lengths"]
    assert c.values.dtype == np.int64 and c.lengths.dtype == np.int32


def test_weighted_map_parse_golden():
    case = G["id_feature_parse_weighted"]["cases"][0]
    c = parse_sparse_column("f", case["input"], is_weighted=True)
    assert c.values.tolist() == case["values"] and c.lengths.tolist() == case["lengths"]
    np.testing.assert_array_equal(c.weights, np.asarray(case["weights"], np.float32))


@pytest.mark.parametrize("name", ["data_parser_nofg", "data_parser_weighted"])
def test_to_batch_golden(name):
    g = G[name]
    cols = {}
    for k, spec in g["columns"].items():
        if k.startswith("click_seq__"):
            cols[k] = parse_sequence_column(k, spec["input"], sequence_delim=spec["sep"], default_value=_default(spec["default"]))
        else:
            cols[k] = parse_sparse_column(k, spec["input"], multival_sep=spec["sep"], default_value=_default(spec["default"]),
                                          is_weighted=bool(spec.get("weighted")))
    kjt = DataParser(g["kjt"]["keys"], sequence_keys=[k for k in cols if k.startswith("click_seq__")]).to_kjt(cols)
    assert kjt.keys() == g["kjt"]["keys"]
    assert kjt.values().tolist() == g["kjt"]["values"]
    assert kjt.lengths().tolist() == g["kjt"]["lengths"]
    assert kjt.stride() == 3
    if "weights" in g["kjt"]:
        np.testing.assert_array_equal(kjt.weights().numpy(), np.asarray(g["kjt"]["weights"], np.float32))
    else:
        assert kjt.weights_or_none() is None


def _random_rows(rng, n, kind):
    rows = []
    for _ in range(n):
        r = rng.integers(0, 6)
        if r == 0:
            rows.append(None)
        elif r == 1 and kind != "int":
            rows.append("" if kind in ("str", "wstr") else ([] if kind == "list" else None))
        else:
            ids = rng.integers(0, 1 << 40, size=rng.integers(1, 5)).tolist()
            if kind == "int":
                rows.append(int(ids[0]))
            elif kind == "str":
                rows.append(S3.join(map(str, ids)))
            elif kind == "wstr":
                rows.append(S3.join(f"{i}:{rng.integers(1, 9) / 4}" for i in ids))
            elif kind == "list":
                rows.append(ids)
            else:  # map
                rows.append({str(i): float(rng.integers(1, 9) / 4) for i in ids})
    return rows


@pytest.mark.parametrize("kind", ["int", "str", "wstr", "list", "map"])
@pytest.mark.parametrize("default", [None, [0], [7, 9]])
def test_sparse_parse_matches_oracle(kind, default):
    if kind == "int" and default is not None and len(default) > 1:
        default = default[:1]
    rng = np.random.default_rng(hash((kind, str(default))) % 2**32)
    rows = _random_rows(rng, 257, kind)
    weighted = kind in ("wstr", "map")
    ev, el, ew = orc.parse_sparse_feature(rows, default, S3, weighted)
    arr = rows
    if kind == "int":
        arr = pa.array(rows, type=pa.int64())
    elif kind == "list":
        arr = pa.array(rows, type=pa.list_(pa.int64()))
    c = parse_sparse_column("f", arr, S3, default, is_weighted=weighted)
    np.testing.assert_array_equal(c.values, ev)
    np.testing.assert_array_equal(c.lengths, el)
    if weighted:
        np.testing.assert_array_equal(c.weights, ew)
    else:
        assert c.weights is None


def test_sequence_and_dense_columns():
    c = parse_sequence_column("s", ["1" + S3 + "2;3", "", None, "4"], default_value=[0])
    assert c.values.tolist() == [1, 2, 3, 0, 0, 4] and c.lengths.tolist() == [2, 1, 1, 1, 1] and c.seq_lengths.tolist() == [2, 1, 1, 1]
    c = parse_sequence_column("s", pa.array([[[1, 2], [3]], [], None], type=pa.list_(pa.list_(pa.int64()))))
    assert c.values.tolist() == [1, 2, 3] and c.lengths.tolist() == [2, 1] and c.seq_lengths.tolist() == [2, 0, 0]
    d = parse_dense_column("d", ["0.5" + S3 + "1.5", "", "2" + S3 + "3"], default_value=[0.0, 0.0])
    np.testing.assert_array_equal(d.values, np.asarray([[0.5, 1.5], [0, 0], [2, 3]], np.float32))
    d = parse_dense_column("d", pa.array([1, None, 3]), default_value=[-1])
    np.testing.assert_array_equal(d.values, np.asarray([[1], [-1], [3]], np.float32))
    with pytest.raises(ValueError):
        parse_dense_column("d", pa.array([1.0, None, 3.0]))
    kt = DataParser([], ["a", "b"]).to_keyed_tensor({"a": parse_dense_column("a", [1.0, 2.0]),
                                                    "b": parse_dense_column("b", ["1" + S3 + "2", "3" + S3 + "4"])})
    assert kt.keys() == ["a", "b"] and kt.length_per_key() == [1, 2] and kt.values().tolist() == [[1, 1, 2], [2, 3, 4]]


def test_use_mask_rows_are_nulled_before_parsing():
    """tzrec/features/id_feature_test.py:190-213 (use_mask, fg_encoded_default_value ""): masked rows
    become null, hence empty bags"""
    from torcheasyrec_amd.data_parser import apply_sample_mask, parse_sparse_column

    col = pa.array(["1\x032", "", None, "3"])
    got = parse_sparse_column("id_feat", apply_sample_mask(col, pa.array([True, False, False, False])))
    assert got.values.tolist() == [3] and got.lengths.tolist() == [0, 0, 0, 1]
    got = parse_sparse_column("id_feat", apply_sample_mask(col, [False, False, False, True]), default_value=[7])
    assert got.values.tolist() == [1, 2, 7, 7, 7] and got.lengths.tolist() == [2, 1, 1, 1]
    ints = apply_sample_mask(pa.array([5, 6, 7], type=pa.int32()), [False, True, False])
    assert parse_sparse_column("x", ints).lengths.tolist() == [1, 0, 1]
    m = pa.array([[("1", 0.5)], [("2", 1.0)]], type=pa.map_(pa.string(), pa.float32()))
    assert apply_sample_mask(m, [True, True]).null_count == 0  # maps are not masked (feature.py:879)


def test_kjt_normalises_dtypes_and_layout():
    """int32 ids (legal in torchrec) and strided views are converted once at construction: the kernels
    reinterpret the buffers as dense int64 / int32 / float32."""
    import pytest
    import torch

    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    vals = torch.arange(12, dtype=torch.int32)[::2]  # int32 AND strided
    k = KeyedJaggedTensor(["a", "b"], vals, torch.tensor([1, 2, 2, 1], dtype=torch.int16), weights=torch.ones(6, dtype=torch.float64))
    assert k.values().dtype == torch.int64 and k.values().is_contiguous() and k.values().tolist() == [0, 2, 4, 6, 8, 10]
    assert k.lengths().dtype == torch.int32 and k.weights().dtype == torch.float32
    with pytest.raises(TypeError):
        KeyedJaggedTensor(["a"], torch.zeros(2), torch.ones(2, dtype=torch.int32))


def test_int32_wire_form_of_a_batch_round_trips():
    """Batch.narrow_ids (the int32 trip across PCIe, tzrec/datasets/utils.py:344-410 moves int64): `.to()` gives back a regular
    KeyedJaggedTensor with the same int64 ids, lengths, weights and hints; ids that do not fit are refused."""
    import torch

    from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, _batch_tensors
    from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor

    vals = torch.tensor([5, 0, (1 << 31) - 1, 7, 7, 2], dtype=torch.int64)
    lens = torch.tensor([2, 0, 1, 3], dtype=torch.int32)
    k = KeyedJaggedTensor(["a", "b"], vals, lens, torch.arange(6, dtype=torch.float32))
    b = Batch({BASE_DATA_GROUP: KeyedTensor(["d"], [1], torch.rand(2, 1))}, {BASE_DATA_GROUP: k}, {"y": torch.tensor([0, 1])})
    w = b.narrow_ids()
    wire = [t for t in _batch_tensors(w) if t.dtype == torch.int32 and t.numel() == 6]
    assert len(wire) == 1 and wire[0].tolist() == vals.tolist()
    back = w.to("cpu").sparse_features[BASE_DATA_GROUP]
    assert isinstance(back, KeyedJaggedTensor) and back.values().dtype == torch.int64
    assert torch.equal(back.values(), vals) and torch.equal(back.lengths(), lens) and torch.equal(back.weights(), k.weights())
    assert back.keys() == ["a", "b"] and back.stride() == 2
    u = KeyedJaggedTensor(["a"], torch.tensor([3, 4]), torch.ones(2, dtype=torch.int32))
    assert u.narrow_ids().to("cpu").uniform_length() == 1
    import pytest

    with pytest.raises(ValueError):
        KeyedJaggedTensor(["a"], torch.tensor([1 << 31]), torch.ones(1, dtype=torch.int32)).narrow_ids()
    with pytest.raises(ValueError):
        KeyedJaggedTensor(["a"], torch.tensor([-1]), torch.ones(1, dtype=torch.int32)).narrow_ids()


def test_ids_per_key_are_counted_on_the_host_side_of_to():
    """`KeyedJaggedTensor.length_per_key()` on a batch in host memory is host arithmetic (no device, no library call), and
    `.to()` carries the cache: the device copy never has to read its offsets back (a synchronising copy, and the end of a
    hipGraph capture)."""
    import torch

    from torcheasyrec_amd.sparse import KeyedJaggedTensor, _perm_tensor

    lengths = torch.tensor([1, 0, 3, 2, 2, 2], dtype=torch.int32)  # 3 keys x 2 samples
    kjt = KeyedJaggedTensor(["a", "b", "c"], torch.arange(10, dtype=torch.int64), lengths)
    assert kjt._length_per_key is None
    assert kjt.length_per_key() == [1, 5, 4]
    moved = kjt.to(torch.device("cpu"))
    assert moved._length_per_key == [1, 5, 4]
    # the device copies of key permutations are made once per (permutation, device)
    assert _perm_tensor((2, 0), torch.device("cpu")) is _perm_tensor((2, 0), torch.device("cpu"))
    assert _perm_tensor((2, 0), torch.device("cpu")).tolist() == [2, 0]


def test_per_key_uniform_hint_survives_permute_and_moves(dev):
    import torch

    """A batch's ONE KeyedJaggedTensor holds sequence keys next to one-id-per-sample keys (tzrec/datasets/data_parser.py:576-585).
    Built on the host, the keys with exactly one id per bag are noted; `permute` to a subset of them is a uniform
    KeyedJaggedTensor again (the pooled collection then takes the kernels' one-id-per-bag forms: no host sync for the id count
    either), a subset with a sequence key is not, and the hint survives `.to()`, `pin_memory`-style copies, `split` and `concat`."""
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    B = 5
    lens = torch.tensor([1] * B + [2, 0, 3, 1, 4] + [1] * B + [1, 1, 0, 1, 1], dtype=torch.int32)
    vals = torch.arange(int(lens.sum()), dtype=torch.int64) * 7 + 3
    k = KeyedJaggedTensor(["a", "seq", "b", "c"], vals, lens)
    assert k.uniform_length() is None and k._uniform_keys == frozenset({"a", "b"})
    kd = k.to(dev)
    assert kd._uniform_keys == frozenset({"a", "b"})
    p = kd.permute([2, 0])
    assert p.uniform_length() == 1 and p.keys() == ["b", "a"]
    assert p.values().cpu().tolist() == vals[B + 10:2 * B + 10].tolist() + vals[:B].tolist()
    assert p.lengths().cpu().tolist() == [1] * (2 * B)
    q = kd.permute([0, 1])
    assert q.uniform_length() is None and q._uniform_keys == frozenset({"a"})
    r = kd.permute([3, 2])  # "c" has an empty bag: not uniform
    assert r.uniform_length() is None
    s0, s1 = k.split([2, 2])
    assert s0._uniform_keys == frozenset({"a"}) and s1._uniform_keys == frozenset({"b"})
    cc = KeyedJaggedTensor.concat([s0, s1])
    assert cc.uniform_length() is None and cc._uniform_keys == frozenset({"a", "b"})
    allone = KeyedJaggedTensor(["x", "y"], torch.arange(2 * B), torch.ones(2 * B, dtype=torch.int32))
    assert allone.uniform_length() == 1  # (the global hint as before)


def test_pooled_collection_takes_the_uniform_prefix_of_a_mixed_batch(dev):
    """EmbeddingBagCollection on a KeyedJaggedTensor that holds a sequence key BEHIND its own one-id-per-bag keys: it runs on the
    uniform view of the keys in front (sparse.uniform_prefix: shared storage) -- same pooled rows and the same fused update as on
    the KeyedJaggedTensor of its keys alone, and the one-id forms are the ones taken (`uniform_length() == 1` reaches the kernels)."""
    import torch

    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    B = 64
    rng = np.random.default_rng(0)

    def build():
        torch.manual_seed(0)
        return EmbeddingBagCollection([EmbeddingBagConfig("ta", 16, 500, ["a"]), EmbeddingBagConfig("tb", 16, 40, ["b"])], device=dev,
                                      optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.1))

    ia, ib = rng.integers(0, 500, B), rng.integers(0, 40, B)
    seq_len = rng.integers(0, 5, B).astype(np.int32)
    seq = rng.integers(0, 100, int(seq_len.sum()))
    mixed = KeyedJaggedTensor(["a", "b", "seq"], torch.from_numpy(np.concatenate([ia, ib, seq])),
                              torch.from_numpy(np.concatenate([np.ones(2 * B, np.int32), seq_len]))).to(dev)
    own = KeyedJaggedTensor(["a", "b"], torch.from_numpy(np.concatenate([ia, ib])), torch.ones(2 * B, dtype=torch.int32)).to(dev)
    assert mixed.uniform_length() is None and own.uniform_length() == 1
    view = mixed.uniform_prefix(["b", "a"])
    assert view is not None and view.uniform_length() == 1 and view.keys() == ["a", "b"]
    assert view.values().data_ptr() == mixed.values().data_ptr()
    assert mixed.uniform_prefix(["a", "seq"]) is None
    g = torch.randn(B, 32, generator=torch.Generator().manual_seed(1)).to(dev)
    res = []
    for kjt in (own, mixed):
        ebc = build()
        out = ebc(kjt).values()
        (out * g).sum().backward()
        res.append((out.detach().cpu(), {n: w.detach().cpu().clone() for n, w in ebc.table_weights().items()}))
    assert torch.equal(res[0][0], res[1][0])
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n
