"""The one-launch backward index plan (csrc/pooled_bwd_cells.hip): its host-built geometry as an object of its own -- every
(table, bucket, chunk) cell belongs to exactly one unit, whatever the tables and the batch size -- and the shapes of unit the
small parity tests never produce (split rows, streamed row groups, many units per table, the demotion to the exact plan)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402

GEO_DT = np.dtype([(n, "<i8") for n in ("n_chunks", "n_units", "n_recs", "n_counters", "n_feats", "max_dim", "ch", "n_positions", "n_bnd",
                                        "off_chunks", "off_units", "off_fstart", "off_fkey", "off_fbo", "off_bnd", "off_recs", "off_rcount",
                                        "off_counters", "off_overflow", "bytes")])
CHUNK_DT = np.dtype([("t", "<i4"), ("nb", "<i4"), ("s", "<i8"), ("e", "<i8"), ("ts", "<i8"), ("mult", "<u8"), ("rows", "<i8"), ("fbase", "<i8"),
                     ("bnd0", "<i4"), ("nbnd", "<i4")])
UNIT_DT = np.dtype([("tb", _lib.TABLE_DT), ("split", "<i4"), ("crel0", "<i4"), ("ncell", "<i4"), ("i0", "<i4"), ("i1", "<i4"), ("b0", "<i4"),
                    ("nrows", "<i4"), ("rec", "<i4"), ("rec0", "<i4"), ("counter", "<i4"), ("feat", "<i4"), ("ts", "<u4")])
assert CHUNK_DT.itemsize == 64 and UNIT_DT.itemsize == 96


def _geometry(rows_list, B, shared=None):
    """(header, chunks, units, boundary list) of the geometry for one key per table (`shared`: table index read by TWO keys)"""
    T = len(rows_list)
    tables = np.zeros(T, dtype=_lib.TABLE_DT)
    feats = []
    for t, r in enumerate(rows_list):
        first = len(feats)
        feats.append((t, len(feats)))
        if shared == t:
            feats.append((t, len(feats)))
        tables[t]["rows"], tables[t]["dim"], tables[t]["w_stride"], tables[t]["first_order"], tables[t]["n_feats"] = r, 16, 16, first, len(feats) - first
    fa = np.zeros(len(feats), dtype=_lib.FEATURE_DT)
    for i, (t, key) in enumerate(feats):
        fa[i]["table"], fa[i]["key"], fa[i]["order"], fa[i]["n_dst"] = t, key, i, 1
    info = (C.c_int64 * 8)()
    L = _lib.lib()
    rc = L.tzr_bwd_cells_geometry(tables.ctypes.data, T, fa.ctypes.data, len(fa), B, 16, None, 0, info)
    if rc != 0:
        return rc, None, None, None
    img = np.zeros(int(info[0]), dtype=np.uint8)
    assert L.tzr_bwd_cells_geometry(tables.ctypes.data, T, fa.ctypes.data, len(fa), B, 16, img.ctypes.data, img.nbytes, info) == 0
    g = img[:GEO_DT.itemsize].view(GEO_DT)[0]
    chunks = img[g["off_chunks"]:g["off_chunks"] + g["n_chunks"] * 64].view(CHUNK_DT)
    units = img[g["off_units"]:g["off_units"] + g["n_units"] * 96].view(UNIT_DT)
    bnd = img[g["off_bnd"]:g["off_bnd"] + g["n_bnd"] * 2].view("<u2")
    assert (info[1], info[2], info[6]) == (g["n_chunks"], g["n_units"], g["off_overflow"])
    return g, chunks, units, bnd


@pytest.fixture
def geo_knobs(emu_path):
    _lib.use_library(emu_path)
    return "default"


@pytest.mark.parametrize("case", ["criteo_65536", "criteo_16384", "tiny_mix", "one_row", "shared_key", "random0", "random1", "random2"])
def test_geometry_covers_every_cell_exactly_once(geo_knobs, case):
    rng = np.random.default_rng(sum(map(ord, case)))
    shared = None
    if case.startswith("criteo"):
        from torcheasyrec_amd.criteo import CRITEO_ROWS
        rows, B = list(CRITEO_ROWS), int(case.split("_")[1])
    elif case == "tiny_mix":
        rows, B = [1, 2, 3, 7, 40, 155, 512, 513, 5000], 30000
    elif case == "one_row":
        rows, B = [1], 60000
    elif case == "shared_key":
        rows, B, shared = [3, 70000, 20], 9000, 1
    else:
        rows = [int(x) for x in np.exp(rng.uniform(0, np.log(5e7), size=int(rng.integers(1, 12))))]
        B = int(rng.integers(1, 150000))
    g, chunks, units, bnd = _geometry(rows, B, shared)
    N = B * (len(rows) + (shared is not None))
    ch_expected = 256 if N <= 512 * 1024 else (512 if N <= 1024 * 1024 else 1024)  # pooled_bwd.h: bwd_pick_ch
    if -(-B * (2 if shared is not None else 1) // ch_expected) > 256:  # more than 256 chunks of one table: not a case for this plan
        assert isinstance(g, int) and g == -4
        return
    assert not isinstance(g, int), g
    ch = int(g["ch"])
    assert ch == ch_expected
    # chunks tile every table's positions
    for t in range(len(rows)):
        ct = chunks[chunks["t"] == t]
        n_t = B * (2 if shared == t else 1)
        assert len(ct) == -(-n_t // ch) and ct["s"][0] == ct["ts"][0] and int((ct["e"] - ct["s"]).sum()) == n_t
        assert bool(np.all(ct["s"][1:] == ct["e"][:-1]))
    # units: per table a grid over (bucket, chunk) with every cell covered once
    first_chunk = {}
    for i, c in enumerate(chunks):
        first_chunk.setdefault(int(c["t"]), i)
    table_of_row = {}
    for c in chunks:
        for i in range(c["bnd0"], c["bnd0"] + c["nbnd"]):
            table_of_row[i] = int(c["t"])
    cover = {t: np.zeros((min(rows[t], 512) if rows[t] <= 512 else 512, int((chunks["t"] == t).sum())), dtype=np.int32) for t in range(len(rows))}
    recs, counters = set(), {}
    for u in units:
        t = table_of_row[int(u["i0"])]
        assert u["tb"]["rows"] == rows[t]
        b_lo, b_hi = int(bnd[u["i0"]]), int(bnd[u["i1"]])
        assert u["b0"] == b_lo and b_hi > b_lo and u["ncell"] >= 1
        assert u["nrows"] == 0 and u["i1"] == u["i0"] + 1
        cover[t][b_lo:b_hi, u["crel0"]:u["crel0"] + u["ncell"]] += 1
        if u["split"] > 0:
            assert b_hi - b_lo == 1 and u["rec"] not in recs and u["rec0"] <= u["rec"] < u["rec0"] + u["split"]
            recs.add(int(u["rec"]))
            counters.setdefault(int(u["counter"]), []).append(int(u["split"]))
        else:
            assert u["ncell"] == cover[t].shape[1]  # only slices of a split row take a chunk range
    for t, cv in cover.items():
        assert bool(np.all(cv == 1)), f"table {t} ({rows[t]} rows): cells covered {np.unique(cv)}"
    assert len(recs) == g["n_recs"] and len(counters) == g["n_counters"] and all(len(v) == v[0] for v in counters.values())
    # sizes: expected lookups of a gathered unit stay 6 sigma under the LDS capacity; the grid fits the budget when it can
    for u in units:
        t = table_of_row[int(u["i0"])]
        n_t = B * (2 if shared == t else 1)
        nb = rows[t] if rows[t] <= 512 else 512
        frac = (int(bnd[u["i1"]]) - int(bnd[u["i0"]])) / nb if rows[t] > 512 else (int(bnd[u["i1"]]) - int(bnd[u["i0"]])) / rows[t]
        if u["nrows"] == 0 and (int(bnd[u["i1"]]) - int(bnd[u["i0"]]) > 1 or u["split"] > 0):
            share = u["ncell"] / cover[t].shape[1]  # (a slice of a split row holds its chunks' share of the row)
            assert n_t * frac * share <= 1076 * (1.0 + 1.0 / max(u["ncell"], 1)) + 1e-6, (rows[t], n_t * frac * share)  # (whole buckets / chunks)
    if case == "criteo_65536":
        assert g["n_units"] + 32 <= 256 * 7, g["n_units"]  # the apply's grid (+ its workers) resident at once


def test_geometry_refuses_what_it_is_not_for(emu_path):
    _lib.use_library(emu_path)
    assert _geometry([1000], 300000)[0] == -4      # more than 256 chunks of one table
    assert _geometry([1000, 5], 0)[0] == -1        # empty batch: invalid


def _run(dev, rows, B, mode, idgen=None, steps=2, kind="adagrad"):
    """weights after `steps` backward passes of one table + a small one, plan `mode`; oracle check on the way"""
    rng = np.random.default_rng(11)
    w0 = [(torch.rand(r, 16, generator=torch.Generator().manual_seed(5 + i)) - 0.5) * 0.2 for i, r in enumerate([rows, 40])]
    cfgs = [EmbeddingBagConfig("t_a", 16, rows, ["a"], "sum", init_fn=lambda t: t.copy_(w0[0])),
            EmbeddingBagConfig("t_s", 16, 40, ["s"], "sum", init_fn=lambda t: t.copy_(w0[1]))]
    ebc = EmbeddingBagCollection(cfgs, device=dev, optimizer=SparseOptimizerConfig(kind=kind, lr=0.05, initial_accumulator_value=0.1))
    ebc.plan_mode = mode
    wr = [w.numpy().copy() for w in w0]
    mr = [np.full_like(w, 0.1) for w in wr]
    opt = orc.SparseOptim(kind=kind, lr=0.05)
    for s in range(steps):
        ids_a = (idgen(rng, rows, B) if idgen else rng.integers(0, rows, size=B)).astype(np.int64)
        ids_s = rng.integers(0, 40, size=B).astype(np.int64)
        kjt = KeyedJaggedTensor(["a", "s"], torch.from_numpy(np.concatenate([ids_a, ids_s])), torch.ones(2 * B, dtype=torch.int32), uniform_length=1)
        g = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32))
        (ebc(kjt.to(dev)).values() * g.to(dev)).sum().backward()
        orc.sparse_update(wr[0], mr[0], ids_a, g[:, :16].numpy(), opt)
        orc.sparse_update(wr[1], mr[1], ids_s, g[:, 16:].numpy(), opt)
    return ebc, wr


@pytest.fixture
def planned(dev):
    assert _lib.lib().tzr_tune(b"bwd_direct", -1) == 0
    yield
    _lib.lib().tzr_tune(b"bwd_direct", 0)


@pytest.mark.parametrize("case", ["split_rows", "row_groups", "sorted_exact", "big_table", "one_row_split"])
def test_cells_unit_shapes_against_the_oracle(dev, planned, case):
    """split_rows: 2 rows x 6 000 lookups each -> each row split over chunk ranges (partial-sum records, last arriver
    combines); row_groups: 40 rows x 750 -> one row per unit; sorted_exact: 300 rows x 67 -> whole rows grouped, gathered and
    sorted; big_table: a bucketed table, 40 units; one_row_split: ONE row, every lookup, split.  fp32 order-of-summation
    tolerance as in test_backward_long_runs."""
    rows, B = {"split_rows": (2, 12000), "row_groups": (40, 30000), "sorted_exact": (300, 20000), "big_table": (3000000, 40000),
               "one_row_split": (1, 9000)}[case]
    ebc, wr = _run(dev, rows, B, "cells", steps=2)
    tol = 5e-4 if rows <= 40 else 2e-5
    np.testing.assert_allclose(ebc.table_weights()["t_a"].detach().cpu().numpy(), wr[0], rtol=tol, atol=1e-6)
    np.testing.assert_allclose(ebc.table_weights()["t_s"].detach().cpu().numpy(), wr[1], rtol=5e-4, atol=1e-6)
    meta = ebc._meta(("a", "s"), ebc._default_layout())
    geo = meta.cells[B]
    assert geo is not None and int(geo.d_overflow.cpu()[0]) == 0  # evenly drawn ids: no unit overflowed


def test_auto_mode_leaves_skewed_ids_to_the_exact_plan(dev, planned):
    """plan_mode "auto": evenly drawn ids stay on the cells plan; ids with a hot row overflow a unit -- handled correctly by the
    worker workgroups of that very launch -- the overflow word moves, and from the next batch on the collection takes the exact
    plan (whose heavy-bucket machinery is made for such ids).  Values against the oracle in every phase."""
    def hot(rng, rows, n):
        ids = rng.integers(0, rows, size=n)
        ids[rng.random(n) < 0.5] = 4242
        return ids

    calls = {"cells": 0, "exact": 0}
    L = _lib.lib()
    orig = {n: getattr(L, n) for n in ("tzr_pooled_bwd_cells_apply", "tzr_pooled_bwd_apply")}
    L.tzr_pooled_bwd_cells_apply = lambda *a: (calls.__setitem__("cells", calls["cells"] + 1), orig["tzr_pooled_bwd_cells_apply"](*a))[1]
    L.tzr_pooled_bwd_apply = lambda *a: (calls.__setitem__("exact", calls["exact"] + 1), orig["tzr_pooled_bwd_apply"](*a))[1]
    try:
        ebc, wr = _run(dev, 100000, 6000, "auto", steps=3)
        assert calls == {"cells": 3, "exact": 0}
        np.testing.assert_allclose(ebc.table_weights()["t_a"].detach().cpu().numpy(), wr[0], rtol=2e-5, atol=1e-6)
        calls.update(cells=0, exact=0)
        ebc, wr = _run(dev, 100000, 6000, "auto", idgen=hot, steps=3)
        assert calls == {"cells": 1, "exact": 2}, calls  # found out after the first batch
        np.testing.assert_allclose(ebc.table_weights()["t_a"].detach().cpu().numpy(), wr[0], rtol=5e-4, atol=1e-6)
        geo = ebc._meta(("a", "s"), ebc._default_layout()).cells[6000]
        assert geo.demoted and int(geo.d_overflow.cpu()[0]) >= 1
    finally:
        for n, f in orig.items():
            setattr(L, n, f)


def test_cells_plan_is_bit_reproducible(dev, planned):
    """same ids, same gradients, twice -> the same bits (split rows, streamed rows, gathered units, and a unit that overflows)"""
    def hot(rng, rows, n):
        ids = rng.integers(0, rows, size=n)
        ids[rng.random(n) < 0.4] = 77
        return ids

    for rows, B, gen in ((2, 12000, None), (100000, 5000, hot), (500, 8000, None)):
        outs = [(_run(dev, rows, B, "cells", idgen=gen, steps=1)[0]) for _ in range(2)]
        a, b = (o.table_weights()["t_a"].detach().cpu() for o in outs)
        assert torch.equal(a, b), (rows, B)
