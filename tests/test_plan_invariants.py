"""The backward index plan (K6) as an object of its own: after `tzr_pooled_bwd_plan`, every table's pairs
{row, lookup position} (`tzr_pooled_bwd_plan_view`) must be a permutation of the table's lookups with equal rows
adjacent and a row's lookups in table-major order (the table's keys in plan order, batch order inside a key) -- whatever the id distribution, bag shape, table sharing
or chunk size.  Randomised over those (seeded), integer work: exact."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402



@pytest.fixture(autouse=True)
def _planned_backward(dev, monkeypatch):
    """these tests inspect the index plan: batches this small would otherwise take the plan-less one-launch backward, and a fused
    plan leaves the units without heavy lookups to the apply's own LDS sort (ks[0] would be incomplete: bwd_no_fuse_sort)"""
    from torcheasyrec_amd import _lib

    assert _lib.lib().tzr_tune(b"bwd_direct", -1) == 0 and _lib.lib().tzr_tune(b"bwd_no_fuse_sort", 1) == 0
    monkeypatch.setenv("TZR_BWD_PLAN", "exact")  # (the four-launch plan is the object inspected here; the one-launch plan: tests/test_cells_plan.py)
    yield
    _lib.lib().tzr_tune(b"bwd_direct", 0)
    _lib.lib().tzr_tune(b"bwd_no_fuse_sort", 0)


def _ids(rng, rows, n, kind):
    if kind == "uniform":
        return rng.integers(0, rows, size=n)
    if kind == "hot":  # a few hot rows + uniform rest
        ids = rng.integers(0, rows, size=n)
        hot = rng.random(n) < rng.uniform(0.2, 0.9)
        ids[hot] = rng.choice(rng.integers(0, rows, size=3), size=int(hot.sum()))
        return ids
    if kind == "narrow":  # a dense run of consecutive rows
        w = int(min(rows, rng.integers(1, 3000)))
        return int(rng.integers(0, rows - w + 1)) + rng.integers(0, w, size=n)
    z = rng.zipf(1.05, size=n).astype(np.int64) - 1  # the bench's clipped Zipf
    return (np.minimum(z, rows - 1) * 2654435761 + 12345) % rows


def _check_plan(ebc, kjt, dev):
    ws = ebc.plan_backward(kjt)
    if dev.type == "cuda":
        torch.cuda.synchronize()
    meta = ebc._meta(kjt.keys(), ebc._default_layout())
    F, T = len(ebc._lookups), len(ebc._configs)
    N, NP = kjt.values().numel(), ebc._n_positions(kjt)
    o8 = (ctypes.c_int64 * 8)()
    assert _lib.lib().tzr_pooled_bwd_plan_view(N, NP, F, T, max(c.embedding_dim for c in ebc._configs), o8) == 0
    host = ws.cpu()
    sorted_pairs = host[o8[0]:o8[0] + 8 * NP].view(torch.int32).view(NP, 2).numpy().astype(np.int64)
    part_pairs = host[o8[1]:o8[1] + 8 * NP].view(torch.int32).view(NP, 2).numpy().astype(np.int64)
    fstart = host[o8[2]:o8[2] + 4 * (F + 1)].view(torch.int32).numpy().astype(np.int64)
    vals = kjt.values().cpu().numpy()
    off = kjt.offsets().cpu().numpy()
    B = kjt.stride()
    keys = list(kjt.keys())
    order = meta.feats_np["order"]
    for t, cfg in enumerate(ebc._configs):
        mine = sorted((int(order[i]), i) for i, lk in enumerate(ebc._lookups) if lk.table == t)
        if not mine:
            continue
        s, e = fstart[mine[0][0]], fstart[mine[-1][0] + 1]
        spans = [(off[keys.index(ebc._lookups[i].key) * B], off[(keys.index(ebc._lookups[i].key) + 1) * B]) for _, i in mine]
        want_src = np.concatenate([np.arange(a, b) for a, b in spans] + [np.zeros(0, np.int64)])
        rank = np.full(N + 1, -1, np.int64)  # the table-major position of a lookup: the order the partition is stable in
        rank[want_src] = np.arange(len(want_src))
        assert e - s == len(want_src), cfg.name
        pairs = (part_pairs if cfg.num_embeddings <= 512 else sorted_pairs)[s:e]
        k, sp = pairs[:, 0], pairs[:, 1]
        assert np.array_equal(np.sort(sp), np.sort(want_src)), f"{cfg.name}: not a permutation of the table's lookups"
        assert np.array_equal(k, np.where((vals[sp] >= 0) & (vals[sp] < cfg.num_embeddings), vals[sp], 0)), f"{cfg.name}: row of a pair != its id"
        if len(k) > 1:
            same = k[1:] == k[:-1]
            assert len(np.unique(k)) == int((~same).sum()) + 1, f"{cfg.name}: equal rows not adjacent"
            r = rank[sp]
            assert bool(np.all(r[1:][same] > r[:-1][same])), f"{cfg.name}: lookups of a row not in table-major order"


@pytest.mark.parametrize("seed", range(24))
def test_plan_invariants_random(dev, seed):
    rng = np.random.default_rng(1000 + seed)
    n_tab = int(rng.integers(1, 6))
    B = int(rng.choice([1, 7, 64, 300, 1100, 2500]))
    jagged = bool(rng.integers(0, 2))
    cfgs, keys, key_rows = [], [], []
    for t in range(n_tab):
        rows = int(rng.choice([1, 3, 200, 512, 513, 700, 5000, 70000, 1 << 20, 3_000_000]))
        feats = [f"k{t}_{j}" for j in range(int(rng.choice([1, 1, 2])))]  # sometimes two keys share the table
        cfgs.append(EmbeddingBagConfig(f"t{t}", 4, rows, feats, trainable=bool(rng.random() > 0.15)))
        keys += feats
        key_rows += [rows] * len(feats)
    perm = rng.permutation(len(keys))  # KJT key order is not the table order
    keys, key_rows = [keys[i] for i in perm], [key_rows[i] for i in perm]
    if all(not c.trainable for c in cfgs):
        cfgs[0].trainable = True
    vals, lens = [], []
    for rows in key_rows:
        L = (rng.poisson(2.0, size=B) if jagged else np.ones(B)).astype(np.int32)
        lens.append(L)
        vals.append(_ids(rng, rows, int(L.sum()), str(rng.choice(["uniform", "hot", "narrow", "zipf"]))))
    kjt = KeyedJaggedTensor(keys, torch.from_numpy(np.concatenate(vals).astype(np.int64)), torch.from_numpy(np.concatenate(lens)),
                            uniform_length=None if jagged else 1)
    ebc = EmbeddingBagCollection(cfgs, device=dev, optimizer=SparseOptimizerConfig(kind="sgd", lr=0.1))
    ch = int(rng.choice([0, 0, 512, 1024]))
    _lib.lib().tzr_tune(b"bwd_ch", ch)
    _lib.lib().tzr_tune(b"bwd_one_wg_heavy", int(seed % 4 == 3))
    try:
        # frozen tables are left out of the plan (ordered last, table -1): check the trainable ones
        live = EmbeddingBagCollection([c for c in cfgs if c.trainable], device=dev, optimizer=SparseOptimizerConfig(kind="sgd", lr=0.1))
        live_keys = [k for k in keys if any(k in c.feature_names for c in cfgs if c.trainable)]
        if set(live_keys) == set(keys):
            _check_plan(ebc, kjt.to(dev), dev)
        else:  # the full collection must still plan without faulting; invariants on the trainable sub-collection
            ebc.plan_backward(kjt.to(dev))
            sel = [i for i, k in enumerate(keys) if k in live_keys]
            kjt2 = KeyedJaggedTensor([keys[i] for i in sel], torch.from_numpy(np.concatenate([vals[i] for i in sel] + [np.zeros(0, np.int64)]).astype(np.int64)),
                                     torch.from_numpy(np.concatenate([lens[i] for i in sel])), uniform_length=None if jagged else 1)
            _check_plan(live, kjt2.to(dev), dev)
    finally:
        _lib.lib().tzr_tune(b"bwd_ch", 0)
        _lib.lib().tzr_tune(b"bwd_one_wg_heavy", 0)
