"""Pin the oracle's index stage against the reference's own golden vectors
(tests/golden/reference_index_vectors.json, transcribed by make_reference_index_vectors.py from
tzrec/features/id_feature_test.py and tzrec/datasets/data_parser_test.py) and check the C-ABI
library exports every symbol include/tzrec_hip.h declares."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.join(os.path.dirname(__file__), "..")
sys.path.insert(0, ROOT)
from oracle import tzrec_oracle as orc  # noqa: E402

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_index_vectors.json")))


@pytest.mark.parametrize("case", G["id_feature_parse"]["cases"])
def test_id_feature_parse_matches_reference(case):
    default = [int(case["default"])] if case["default"] != "" else None
    v, l, w = orc.parse_sparse_feature(case["input"], default)
    assert v.dtype == np.int64 and v.tolist() == case["values"]
    assert l.tolist() == case["lengths"]
    assert w is None


def test_weighted_map_parse_matches_reference():
    case = G["id_feature_parse_weighted"]["cases"][0]
    v, l, w = orc.parse_sparse_feature(case["input"])
    assert v.tolist() == case["values"] and l.tolist() == case["lengths"]
    np.testing.assert_allclose(w, case["weights"])


@pytest.mark.parametrize("name", ["data_parser_nofg", "data_parser_weighted"])
def test_kjt_assembly_matches_reference(name):
    case = G[name]
    vals, lens, wts = [], [], []
    for k in case["kjt"]["keys"]:
        col = case["columns"][k]
        v, l, w = orc.parse_sparse_feature(col["input"], col["default"], col["sep"], col.get("weighted", False))
        vals.append(v)
        lens.append(l)
        wts.append(w)
    kjt = orc.to_kjt(case["kjt"]["keys"], vals, lens, wts)
    assert kjt["values"].tolist() == case["kjt"]["values"]
    assert kjt["lengths"].tolist() == case["kjt"]["lengths"]
    assert kjt["stride"] == 3
    if "weights" in case["kjt"]:
        np.testing.assert_allclose(kjt["weights"], case["kjt"]["weights"], rtol=1e-6)
    else:
        assert kjt["weights"] is None
    # offsets of the pinned KJT (torchrec: [0] + cumsum(lengths))
    off = orc.lengths_to_offsets(kjt["lengths"])
    assert off[-1] == len(case["kjt"]["values"])


def test_kjt_container_matches_reference_fields(emu_path):
    """The package's KeyedJaggedTensor built from the pinned vectors exposes torchrec's fields."""
    import torch

    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    _lib.use_library(emu_path)
    k = G["data_parser_nofg"]["kjt"]
    kjt = KeyedJaggedTensor.from_lengths_sync(k["keys"], torch.tensor(k["values"]), torch.tensor(k["lengths"], dtype=torch.int32))
    assert kjt.stride() == 3 and kjt.keys() == k["keys"]
    assert kjt.offsets().tolist() == [0, 1, 2, 3, 5, 5, 6, 9, 10, 10]
    assert kjt.length_per_key() == [3, 3, 4]


def test_shape_fixtures():
    import torch

    f = G["shape_fixtures"]
    x = torch.randn(f["interaction"]["batch"], f["interaction"]["feature_num"], 16)
    assert list(orc.dot_interaction(x).shape) == f["interaction"]["output"]
    assert list(orc.fm(torch.randn(*f["fm"]["input"])).shape) == f["fm"]["output"]


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tzrec_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tzr_[a-z0-9_]+)\s*\(", text)))


def test_abi_library_exports_every_declared_symbol():
    """The hipcc-built library loads (no GPU needed for dlopen) and exports the whole header."""
    from torcheasyrec_amd import _build, _lib

    path = _build.build()
    handle = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for sym in declared:
        assert hasattr(handle, sym), f"{sym} declared in include/tzrec_hip.h but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == set(declared), set(_lib.EXPORTED_SYMBOLS) ^ set(declared)
    handle.tzr_backend.restype = ctypes.c_char_p
    assert handle.tzr_backend() == b"hip-gfx950"


def test_product_path_has_no_cpu_fallback(emu_path):
    """With the real library selected, CPU tensors are refused loudly."""
    import torch

    from torcheasyrec_amd import _build, _lib

    _lib.use_library(_build.build())
    try:
        with pytest.raises(_lib.TzrError):
            _lib.ptr(torch.zeros(4))
    finally:
        _lib.use_library(emu_path)


@pytest.mark.parametrize("kind,mode", [("adagrad", "sum"), ("adagrad", "mean"), ("sgd", "sum")])
def test_oracle_sparse_update_matches_torch_sparse_optimizers(kind, mode):
    """The fused-backward restatement (lookup_grads + sparse_update) against an independent
    implementation that IS installed: nn.EmbeddingBag(sparse=True) + torch.optim.Adagrad / SGD on the
    coalesced sparse gradient -- the 'duplicates summed, then ONE update per row' semantics fbgemm's
    exact optimizers document.  (Row-wise Adagrad has no torch counterpart: still unpinned.)"""
    rng = np.random.default_rng(5)
    rows, D, B, lr, eps, acc0 = 37, 8, 24, 0.05, 1e-8, 0.1
    bag = torch.nn.EmbeddingBag(rows, D, mode=mode, sparse=True, include_last_offset=True)
    w = bag.weight.detach().numpy().copy()
    m = np.full((rows, D), acc0 if kind == "adagrad" else 0.0, dtype=np.float32)
    if kind == "adagrad":
        opt = torch.optim.Adagrad(bag.parameters(), lr=lr, eps=eps, initial_accumulator_value=acc0)
    else:
        opt = torch.optim.SGD(bag.parameters(), lr=lr)
    cfg = orc.SparseOptim(kind=kind, lr=lr, eps=eps, initial_accumulator_value=acc0)
    for _ in range(4):
        lengths = rng.integers(0, 5, size=B)
        ids = rng.integers(0, 6, size=int(lengths.sum()))  # 6 hot rows: many duplicates
        ids[::3] = rng.integers(0, rows, size=len(ids[::3]))
        Gy = rng.standard_normal((B, D)).astype(np.float32)
        off = np.concatenate([[0], np.cumsum(lengths)])
        opt.zero_grad()
        out = bag(torch.from_numpy(ids), torch.from_numpy(off))
        (out * torch.from_numpy(Gy)).sum().backward()
        opt.step()
        g = orc.lookup_grads([Gy], lengths, B, [mode])
        orc.sparse_update(w, m, ids, g, cfg)
        np.testing.assert_allclose(w, bag.weight.detach().numpy(), rtol=2e-6, atol=2e-7)
    if kind == "adagrad":
        np.testing.assert_allclose(m, opt.state[bag.weight]["sum"].numpy(), rtol=2e-6, atol=1e-7)


def test_kjt_host_views(emu_path):
    """torchrec KJT conveniences tzrec calls on the batch (to_dict / [] / split / concat /
    offset_per_key / from_offsets_sync): plain slices of the key-major layout"""
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    _lib.use_library(emu_path)
    lengths = torch.tensor([2, 0, 1, 1, 3, 0, 0, 2, 1], dtype=torch.int32)  # 3 keys x 3 samples
    values = torch.arange(10, dtype=torch.int64) + 100
    w = torch.arange(10, dtype=torch.float32) / 10
    kjt = KeyedJaggedTensor.from_lengths_sync(["a", "b", "c"], values, lengths, w)
    assert kjt.length_per_key() == [3, 4, 3] and kjt.offset_per_key() == [0, 3, 7, 10]
    d = kjt.to_dict()
    assert list(d) == ["a", "b", "c"]
    assert d["b"].values().tolist() == [103, 104, 105, 106] and d["b"].lengths().tolist() == [1, 3, 0]
    assert d["b"].offsets().tolist() == [0, 1, 4, 4] and d["b"].weights_or_none().tolist() == w[3:7].tolist()
    assert kjt["c"].values().tolist() == [107, 108, 109]
    ab, c = kjt.split([2, 1])
    assert ab.keys() == ["a", "b"] and ab.values().tolist() == list(range(100, 107)) and ab.stride() == 3
    assert c.keys() == ["c"] and c.lengths().tolist() == [0, 2, 1] and c.offsets().tolist() == [0, 0, 2, 3]
    back = KeyedJaggedTensor.concat([c, ab])
    assert back.keys() == ["c", "a", "b"] and back.values().tolist() == [107, 108, 109] + list(range(100, 107))
    assert back.lengths().tolist() == [0, 2, 1, 2, 0, 1, 1, 3, 0] and back.weights().tolist() == torch.cat([w[7:], w[:7]]).tolist()
    same = KeyedJaggedTensor.from_offsets_sync(["a", "b", "c"], values, kjt.offsets(), w)
    assert same.lengths().tolist() == lengths.tolist() and same.stride() == 3
    with pytest.raises(ValueError):
        kjt.split([1, 1])
    with pytest.raises(ValueError):
        KeyedJaggedTensor.concat([ab, KeyedJaggedTensor(["z"], values[:2], torch.tensor([1, 1], dtype=torch.int32))])
    assert KeyedJaggedTensor.empty().keys() == []


def test_oracle_sparse_adam_follows_torch_sparse_adam():
    """The Adam restatement against torch.optim.SparseAdam over several steps with duplicates.  The two
    differ only in where eps sits (fbgemm: sqrt(v / (1 - b2^t)) + eps; torch: sqrt(v) / sqrt(1 - b2^t)
    + eps ... i.e. eps scaled by sqrt(1 - b2^t)), invisible at eps = 1e-8 for gradients of order 1."""
    rng = np.random.default_rng(9)
    rows, D, B, lr = 29, 4, 16, 0.01
    bag = torch.nn.EmbeddingBag(rows, D, mode="sum", sparse=True, include_last_offset=True)
    opt = torch.optim.SparseAdam(bag.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8)
    w = bag.weight.detach().numpy().copy()
    m = np.zeros((rows, 2 * D), np.float32)
    cfg = orc.SparseOptim(kind="adam", lr=lr, eps=1e-8)
    for step in range(1, 6):
        lengths = rng.integers(0, 4, size=B)
        ids = rng.integers(0, 7, size=int(lengths.sum()))
        Gy = rng.standard_normal((B, D)).astype(np.float32)
        off = np.concatenate([[0], np.cumsum(lengths)])
        opt.zero_grad()
        (bag(torch.from_numpy(ids), torch.from_numpy(off)) * torch.from_numpy(Gy)).sum().backward()
        opt.step()
        orc.sparse_update(w, m, ids, orc.lookup_grads([Gy], lengths, B, ["sum"]), cfg, step=step)
        np.testing.assert_allclose(w, bag.weight.detach().numpy(), rtol=2e-5, atol=2e-6)
    st = opt.state[bag.weight]
    np.testing.assert_allclose(m[:, :D], st["exp_avg"].numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(m[:, D:], st["exp_avg_sq"].numpy(), rtol=1e-4, atol=1e-8)  # v += (1-b2)(g^2 - v) vs b2 v + (1-b2) g^2
