"""Parity of the pooled lookup forward (K5+K8) and fused backward (K6+K7) against the oracle.

Tolerances (north_star): index stage bit-exact; fp32 pooled embeddings within 1e-5 relative.
L=1 sum pooling is a copy and must be bit-exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402

RTOL = 1e-5


def _direct_forced():
    from torcheasyrec_amd import _lib
    return bool(_lib.lib().tzr_pooled_bwd_direct_supported(1 << 30, 1, 1, 1, 0))  # only the forced knob takes a batch this size


def _make_tables(spec, seed=0):
    """spec: list of (name, rows, dim, pooling, [features])"""
    cfgs, inits = [], {}
    g = torch.Generator().manual_seed(seed)
    for name, rows, dim, pooling, feats in spec:
        w = (torch.rand(rows, dim, generator=g) - 0.5) * 0.2
        inits[name] = w
        cfgs.append(EmbeddingBagConfig(name, dim, rows, list(feats), pooling,
                                       init_fn=lambda t, w=w: t.copy_(w)))
    return cfgs, inits


def _make_kjt(keys, rows_per_key, B, rng, mode="uniform1", weighted=False, idgen=None):
    """idgen(rng, rows, n) -> int64[n] overrides the uniform id draw (hot rows, narrow ranges)."""
    vals, lens = [], []
    for k, rows in zip(keys, rows_per_key):
        if mode == "uniform1":
            L = np.ones(B, dtype=np.int32)
        elif mode == "jagged":
            L = rng.poisson(2.0, size=B).astype(np.int32)
            L[rng.integers(0, B, size=max(B // 8, 1))] = 0
        elif mode == "long":
            L = rng.integers(0, 70, size=B).astype(np.int32)
        else:
            raise ValueError(mode)
        lens.append(L)
        if idgen is not None:
            vals.append(np.asarray(idgen(rng, rows, int(L.sum())), dtype=np.int64))
        else:
            vals.append(rng.integers(0, rows, size=int(L.sum())).astype(np.int64))
    values = torch.from_numpy(np.concatenate(vals))
    lengths = torch.from_numpy(np.concatenate(lens))
    weights = None
    if weighted:
        weights = torch.from_numpy(rng.uniform(0.5, 1.5, size=values.numel()).astype(np.float32))
    return KeyedJaggedTensor(keys, values, lengths, weights)


def _oracle_blocks(ebc_spec, inits, kjt_cpu, lookups, req_grad=False):
    B = kjt_cpu.stride()
    keys = kjt_cpu.keys()
    tabs = {n: inits[n].clone().requires_grad_(req_grad) for n in inits}
    name_of = {}
    pool_of = {}
    for name, rows, dim, pooling, feats in ebc_spec:
        for f in feats:
            name_of.setdefault(f, []).append(name)
            pool_of[name] = pooling
    blocks = {}
    # one block per (table, feature) in table-then-feature order
    v, l, w = _reorder_kjt(kjt_cpu, [f for (f, t) in lookups])
    all_blocks = orc.pooled_lookup(
        [tabs[t] for (f, t) in lookups], [pool_of[t] for (f, t) in lookups], v, l, B, w
    )
    for (f, t), blk in zip(lookups, all_blocks):
        key = f if len(name_of[f]) == 1 else f"{f}@{t}"
        blocks[key] = blk
    return blocks, tabs


def _reorder_kjt(kjt, feat_list):
    """values/lengths(/weights) of the listed keys concatenated key-major (CPU)."""
    B = kjt.stride()
    off = orc.lengths_to_offsets(kjt.lengths().numpy())
    keys = kjt.keys()
    vs, ls, ws = [], [], []
    for f in feat_list:
        i = keys.index(f)
        s, e = off[i * B], off[(i + 1) * B]
        vs.append(kjt.values()[s:e])
        ls.append(kjt.lengths()[i * B:(i + 1) * B])
        if kjt.weights_or_none() is not None:
            ws.append(kjt.weights_or_none()[s:e])
    return torch.cat(vs), torch.cat(ls), (torch.cat(ws) if ws else None)


def _lookup_list(spec):
    return [(f, name) for name, rows, dim, pooling, feats in spec for f in feats]


SPEC_CRITEO_SMALL = [
    ("t_big", 5000, 16, "sum", ["c0"]),
    ("t_mid", 300, 16, "sum", ["c1"]),
    ("t_tiny", 3, 16, "sum", ["c2"]),
    ("t_four", 4, 16, "sum", ["c3"]),
]


@pytest.mark.parametrize("B", [1, 7, 64, 300])
def test_forward_uniform1_bitexact(dev, B):
    rng = np.random.default_rng(B)
    cfgs, inits = _make_tables(SPEC_CRITEO_SMALL)
    ebc = EmbeddingBagCollection(cfgs, device=dev)
    keys = ["c0", "c1", "c2", "c3"]
    kjt = _make_kjt(keys, [5000, 300, 3, 4], B, rng)
    assert kjt.uniform_length() == 1
    out = ebc(kjt.to(dev))
    blocks, _ = _oracle_blocks(SPEC_CRITEO_SMALL, inits, kjt, _lookup_list(SPEC_CRITEO_SMALL))
    ref = torch.cat([blocks[k] for k in out.keys()], dim=1)
    assert out.values().shape == (B, 64)
    assert torch.equal(out.values().cpu(), ref)  # a copy: bit-exact


@pytest.mark.parametrize("tile_b", [0, 8, 32, 64])
def test_forward_uniform1_lds_ids_kernel(dev, tile_b):
    """The one-id-per-bag kernel that stages a tile's ids in LDS (tzr_pooled_fwd_u1_kernel): more slots
    than one workgroup row holds (3 rows of 128), mixed dims, a key read through two tables and copied
    into two groups, a key of the batch nobody reads, out-of-range ids (-> row 0), sub-tiled id passes
    (groups x tile > 1024) and a ragged last tile -- bit-exact against the oracle and against the
    general kernel (tzr_tune fwd_variant = 1)."""
    from torcheasyrec_amd import _lib

    rng = np.random.default_rng(100 + tile_b)
    B = 77
    dims = [16, 4, 8, 32]
    spec, keys, rows = [], [], []
    for i in range(44):
        r = 50 if i == 3 else int(rng.integers(1, 400))
        spec.append((f"t{i}", r, dims[i % 4], "sum", [f"k{i}"]))
        keys.append(f"k{i}")
        rows.append(r)
    spec.append(("t_second", 50, 16, "sum", ["k3"]))  # k3 read through two tables
    keys.insert(5, "nobody")
    rows.insert(5, 9)
    cfgs, inits = _make_tables(spec)
    feats = [f"k{i}" if i != 3 else "k3@t3" for i in range(44)] + ["k3@t_second"]
    groups = {"all": feats, "some": [feats[7], feats[44], feats[0], feats[3]]}
    ebc = EmbeddingBagCollection(cfgs, device=dev, groups=groups)
    kjt = _make_kjt(keys, rows, B, rng)
    v = kjt.values().clone()
    v[rng.integers(0, v.numel(), size=9)] = 10 ** 9  # out of range: read row 0
    v[0] = -1
    kjt = KeyedJaggedTensor(keys, v, kjt.lengths(), uniform_length=1)
    L = _lib.lib()
    assert L.tzr_tune(b"fwd_tile_b", tile_b) == 0
    assert L.tzr_tune(b"fwd_variant", 2) == 0  # the LDS-ids kernel whatever the batch size
    out = {g: t.cpu() for g, t in ebc.forward_grouped(kjt.to(dev)).items()}
    assert L.tzr_tune(b"fwd_variant", 1) == 0
    try:
        gen = {g: t.cpu() for g, t in ebc.forward_grouped(kjt.to(dev)).items()}
    finally:
        L.tzr_tune(b"fwd_variant", 0)
    safe = v.clone()
    for i, r in enumerate(rows):
        seg = safe[i * B:(i + 1) * B]
        seg[(seg < 0) | (seg >= r)] = 0
    blocks, _ = _oracle_blocks(spec, inits, KeyedJaggedTensor(keys, safe, kjt.lengths(), uniform_length=1), _lookup_list(spec))
    ref = orc.regroup(blocks, groups)
    for g in groups:
        assert torch.equal(out[g], ref[g]), g
        assert torch.equal(gen[g], ref[g]), g


def test_forward_uniform1_single_slot(dev):
    """One table of dim 4 = one float4 slot per sample (the k / n_slots quotient of the LDS-ids kernel at n = 1)."""
    from torcheasyrec_amd import _lib

    assert _lib.lib().tzr_tune(b"fwd_variant", 2) == 0
    rng = np.random.default_rng(2)
    spec = [("w", 23, 4, "sum", ["k"])]
    cfgs, inits = _make_tables(spec)
    ebc = EmbeddingBagCollection(cfgs, device=dev)
    kjt = _make_kjt(["k"], [23], 1000, rng)
    out = ebc(kjt.to(dev))
    assert torch.equal(out.values().cpu(), inits["w"][kjt.values()])


SPEC_MIXED = [
    ("u_emb", 1000, 16, "sum", ["user", "user_hist"]),  # shared table, two keys
    ("i_emb", 57, 8, "mean", ["item"]),
    ("w_emb", 1000, 4, "sum", ["wide_user"]),
    ("big_d", 40, 32, "sum", ["ctx"]),
]


@pytest.mark.parametrize("mode,weighted", [("jagged", False), ("jagged", True), ("long", False)])
def test_forward_jagged(dev, mode, weighted):
    rng = np.random.default_rng(11)
    B = 37
    cfgs, inits = _make_tables(SPEC_MIXED)
    ebc = EmbeddingBagCollection(cfgs, device=dev)
    keys = ["ctx", "item", "unused_key", "user", "user_hist", "wide_user"]
    rows = [40, 57, 10, 1000, 1000, 1000]
    kjt = _make_kjt(keys, rows, B, rng, mode=mode, weighted=weighted)
    out = ebc(kjt.to(dev))
    blocks, _ = _oracle_blocks(SPEC_MIXED, inits, kjt, _lookup_list(SPEC_MIXED))
    assert out.keys() == ["user", "user_hist", "item", "wide_user", "ctx"]
    ref = torch.cat([blocks[k] for k in out.keys()], dim=1)
    torch.testing.assert_close(out.values().cpu(), ref, rtol=RTOL, atol=1e-6)


def test_forward_grouped_deepfm_layout(dev):
    """DeepFM: key read through two tables (wide dim 4 + deep dim 16), deep block copied into the
    `fm` and `deep` groups (/root/reference/tzrec/modules/embedding.py:744-786,972-976)."""
    rng = np.random.default_rng(5)
    B = 50
    spec = [
        ("a_emb", 100, 16, "sum", ["a"]),
        ("b_emb", 7, 16, "sum", ["b"]),
        ("a_emb_wide", 100, 4, "sum", ["a"]),
        ("b_emb_wide", 7, 4, "sum", ["b"]),
    ]
    cfgs, inits = _make_tables(spec)
    groups = {
        "wide": ["a@a_emb_wide", "b@b_emb_wide"],
        "fm": ["a@a_emb", "b@b_emb"],
        "deep": ["b@b_emb", "a@a_emb"],
    }
    ebc = EmbeddingBagCollection(cfgs, device=dev, groups=groups)
    kjt = _make_kjt(["a", "b"], [100, 7], B, rng)
    out = ebc.forward_grouped(kjt.to(dev))
    blocks, _ = _oracle_blocks(spec, inits, kjt, _lookup_list(spec))
    ref = orc.regroup(blocks, groups)
    for g in groups:
        assert torch.equal(out[g].cpu(), ref[g]), g


def _run_backward_case(dev, spec, keys, rows, B, mode, weighted, opt_cfg, groups=None, steps=2, seed=3,
                       rtol=2e-5, idgen=None):
    rng = np.random.default_rng(seed)
    cfgs, inits = _make_tables(spec)
    ebc = EmbeddingBagCollection(cfgs, device=dev, optimizer=opt_cfg, groups=groups)
    lookups = _lookup_list(spec)
    # oracle state
    w_ref = {n: inits[n].numpy().copy() for n in inits}
    if opt_cfg.kind == "adagrad":
        m_ref = {n: np.full_like(w_ref[n], opt_cfg.initial_accumulator_value) for n in w_ref}
    elif opt_cfg.kind == "rowwise_adagrad":
        m_ref = {n: np.zeros(w_ref[n].shape[0], np.float32) for n in w_ref}
    elif opt_cfg.kind == "adam":
        m_ref = {n: np.zeros((w_ref[n].shape[0], 2 * w_ref[n].shape[1]), np.float32) for n in w_ref}
    else:
        m_ref = {n: None for n in w_ref}
    oopt = orc.SparseOptim(kind=opt_cfg.kind, lr=opt_cfg.lr, eps=opt_cfg.eps,
                           weight_decay=opt_cfg.weight_decay, weight_decay_mode=opt_cfg.weight_decay_mode,
                           gradient_clipping=opt_cfg.gradient_clipping, max_gradient=opt_cfg.max_gradient,
                           beta1=opt_cfg.beta1, beta2=opt_cfg.beta2)
    pool_of = {name: pooling for name, _, _, pooling, _ in spec}
    for step in range(steps):
        kjt = _make_kjt(keys, rows, B, rng, mode=mode, weighted=weighted, idgen=idgen)
        kd = kjt.to(dev)
        if groups is None:
            out = ebc(kd).values()
            outs = {"__all__": out}
            layout = {"__all__": [f if sum(1 for (ff, _) in lookups if ff == f) == 1 else f"{f}@{t}" for f, t in lookups]}
        else:
            outs = ebc.forward_grouped(kd)
            layout = groups
        gens = {g: torch.from_numpy(rng.standard_normal(tuple(outs[g].shape)).astype(np.float32)) for g in outs}
        loss = sum((outs[g] * gens[g].to(dev)).sum() for g in outs)
        loss.backward()
        # oracle: per lookup block gradient = sum over groups containing it
        name_count = {}
        for f, t in lookups:
            name_count[f] = name_count.get(f, 0) + 1
        off = orc.lengths_to_offsets(kjt.lengths().numpy())
        for f, t in lookups:
            ok = f if name_count[f] == 1 else f"{f}@{t}"
            D = w_ref[t].shape[1]
            gblk = np.zeros((B, D), np.float32)
            for g, oks in layout.items():
                col = 0
                for k2 in oks:
                    d2 = [w_ref[tt].shape[1] for (ff, tt) in lookups if (ff if name_count[ff] == 1 else f"{ff}@{tt}") == k2][0]
                    if k2 == ok:
                        gblk = gblk + gens[g].numpy()[:, col:col + D]
                    col += d2
            ki = keys.index(f)
            s, e = off[ki * B], off[(ki + 1) * B]
            L = kjt.lengths().numpy()[ki * B:(ki + 1) * B]
            psw = kjt.weights_or_none().numpy()[s:e] if weighted else None
            lg = orc.lookup_grads([gblk], L, B, [pool_of[t]], psw)
            # features sharing a table must be applied together: collect
            w_ref.setdefault("__pending__", {}).setdefault(t, []).append((kjt.values().numpy()[s:e], lg))
        pend = w_ref.pop("__pending__")
        for t, items in pend.items():
            ids = np.concatenate([i for i, _ in items])
            gr = np.concatenate([g for _, g in items], axis=0)
            orc.sparse_update(w_ref[t], m_ref[t], ids, gr, oopt, step=step + 1)
    for n in inits:
        got = ebc.table_weights()[n].detach().cpu().numpy()
        np.testing.assert_allclose(got, w_ref[n], rtol=rtol, atol=1e-7, err_msg=f"weights of {n}")
        if m_ref[n] is not None:
            gotm = ebc.table_states()[n].detach().cpu().numpy()
            np.testing.assert_allclose(gotm, m_ref[n], rtol=rtol, atol=1e-7, err_msg=f"state of {n}")


@pytest.mark.parametrize("kind", ["adagrad", "rowwise_adagrad", "sgd"])
def test_backward_uniform1(dev, kind, bwd_path):
    opt = SparseOptimizerConfig(kind=kind, lr=0.05)
    _run_backward_case(dev, SPEC_CRITEO_SMALL, ["c0", "c1", "c2", "c3"], [5000, 300, 3, 4], 200,
                       "uniform1", False, opt)
    assert (bwd_path["direct"] > 0) == (bwd_path is not None and _direct_forced())
    assert (bwd_path["cells"] > 0, bwd_path["exact"] > 0) == (bwd_path["path"] == "cells", bwd_path["path"] == "planned")


@pytest.mark.parametrize("mode,weighted,wd", [("uniform1", False, 0.0), ("jagged", True, 0.01)])
def test_backward_sparse_adam(dev, mode, weighted, wd, bwd_path):
    """adam_optimizer (protos/optimizer.proto:89-96): state [exp_avg | exp_avg_sq], one step counter on
    the device advanced per backward, bias correction as fbgemm's split Adam; 3 steps so the
    correction terms move; clipping and weight decay in the second case"""
    opt = SparseOptimizerConfig(kind="adam", lr=0.01, beta1=0.8, beta2=0.95, weight_decay=wd,
                                gradient_clipping=weighted, max_gradient=0.9)
    _run_backward_case(dev, SPEC_CRITEO_SMALL, ["c0", "c1", "c2", "c3"], [5000, 300, 3, 4], 64, mode, weighted, opt,
                       steps=3, rtol=5e-5)


def test_backward_long_runs(dev, bwd_path):
    """1- and 3-row tables with thousands of lookups: the long-run piece path.

    ~900 to 2,600 random-sign gradients are summed per row; the kernel reduces them with a fixed tree, the
    oracle sequentially, so the comparison carries fp32 order-of-summation noise (cancellation
    amplifies it on g, and m = g*g doubles it): tolerance 5e-4 here, 2e-5 everywhere else.
    """
    opt = SparseOptimizerConfig(kind="adagrad", lr=0.05)
    # 2600 lookups per table: the 1-row table is ONE run over three 1024-position chunks (leading piece,
    # whole-chunk piece, trailing piece), the 3-row table's runs cross chunk and wave-range boundaries
    spec = [("t_one", 1, 16, "sum", ["c0"]), ("t_tiny", 3, 16, "sum", ["c1"])]
    _run_backward_case(dev, spec, ["c0", "c1"], [1, 3], 2600, "uniform1", False, opt, steps=1, rtol=5e-4)
    # (cells plan: both tables' rows are SPLIT over chunk ranges -- partial-sum records, the last unit to arrive combines)
    assert (bwd_path["cells"] > 0) == (bwd_path["path"] == "cells")


def _hot(frac, hot_ids):
    """a fraction of the lookups hits a few hot rows (Zipf head / default id), the rest is uniform"""
    def gen(rng, rows, n):
        ids = rng.integers(0, rows, size=n)
        hot = rng.random(n) < frac
        ids[hot] = rng.choice(np.asarray(hot_ids) % rows, size=int(hot.sum()))
        return ids
    return gen


def _narrow(lo, width):
    """every id inside [lo, lo + width): many distinct rows in one or two buckets"""
    return lambda rng, rows, n: lo % rows + rng.integers(0, min(width, rows), size=n)


def _zipf_clipped(rng, rows, n):
    """the bench's --dist zipf (torcheasyrec_amd/criteo.py): Zipf(1.05) ranks clipped to the table, so the
    last rank collects the whole tail (40-70 % of the lookups on ONE row), scattered multiplicatively"""
    z = rng.zipf(1.05, size=n).astype(np.int64) - 1
    return (np.minimum(z, rows - 1) * 2654435761 + 12345) % rows


# (rows, B, id generator, also on the CPU lane emulator): what the plan's bucket partition, the
# unit-local sort and the heavy kernel see.  The emulator runs the cheap half (one OS thread per lane).
PLAN_CASES = {
    "just_over_exact": (600, 700, None, True),                     # 513..1024 rows: 1-2 row ids per bucket
    "light_units": (70000, 1300, None, True),                      # ~4 lookups per bucket, grouped unit sort
    "three_pass": (1 << 22, 1100, None, True),                    # units spanning ~21 bits of row id
    "wide_rows": (40_000_000, 1500, None, False),                  # ~25 bits
    "hot_one_tile": (70000, 1500, _hot(0.45, [31337]), True),     # one heavy bucket < one heavy tile, rest light
    "hot_t1_tiles": (70000, 2200, _hot(0.6, [31337]), True),       # heavy bucket of <= 512 row ids, 2 parallel tiles
    "zipf_mid_table": (12973, 2600, _zipf_clipped, True),          # ~3000 lookups on one row + Zipf head: tiles + mixed units
    "zipf_big_table": (3067956, 5000, _zipf_clipped, True),       # wide buckets: hot-row tiles
    "hot_two_in_bucket": (200000, 1800, _hot(0.6, [5000, 5001, 5003]), True),  # wide heavy bucket with 3 hot rows
    "hot_multi_tile": (1 << 20, 2000, _hot(0.75, [777777]), True), # wide heavy bucket of ~2500 = 3 hot-row tiles
    "hot_many_tiles": (1 << 20, 40000, _hot(0.5, [777777, 12]), True),  # two heavy buckets of ~10000
    "many_chunks": (70000, 70000, None, True),                     # 273 chunks per table: the scan launch works in 3 slices
    "narrow_dense": (1 << 22, 3000, _narrow(123456, 3000), True), # one wide bucket, ~1900 distinct rows: LSD fallback
    "narrow_fallback": (1 << 20, 1700, _narrow(123456, 1600), True),  # the same at emulator size
    "narrow_two_buckets": (1 << 22, 1400, _narrow(8192 * 3 - 300, 700), True),  # straddles a bucket boundary
}


@pytest.mark.parametrize("case,ch", [(c, 0) for c in sorted(PLAN_CASES)] +
                         [(c, ch) for c in ("light_units", "hot_multi_tile", "narrow_two_buckets", "wide_rows", "hot_many_tiles",
                                            "zipf_mid_table") for ch in (512, 1024)])
def test_backward_plan_shapes(dev, case, ch, bwd_path):
    """ch = positions per chunk of the plan (pooled_bwd.h: bwd_pick_ch); 0 = by problem size (256 here)"""
    from torcheasyrec_amd import _lib
    _lib.lib().tzr_tune(b"bwd_ch", ch)
    try:
        _plan_shape_case(dev, case, ch)
    finally:
        _lib.lib().tzr_tune(b"bwd_ch", 0)


def _plan_shape_case(dev, case, ch=0):
    """The backward plan on id distributions that exercise each of its paths (pooled_bwd.hip): exact
    vs bucketed tables, units made of light buckets, heavy buckets (hot rows) sorted by the heavy
    kernel in one and in several tiles, units mixing slices of heavy buckets with light ones."""
    rows, B, idgen, on_emu = PLAN_CASES[case]
    if dev.type == "cpu" and not on_emu:
        pytest.skip("GPU-only size")
    opt = SparseOptimizerConfig(kind="adagrad", lr=0.05, initial_accumulator_value=0.1)

    spec = [("t_a", rows, 16 if rows < (1 << 20) else 4, "sum", ["c0"]), ("t_small", 40, 16, "sum", ["c1"])]
    # hot rows sum thousands of random-sign gradients: order-of-summation noise as in test_backward_long_runs
    rtol = 5e-4 if case.startswith(("hot", "zipf", "many")) else 2e-5  # (many_chunks: 1 750 gradients per row of the 40-row table)
    _run_backward_case(dev, spec, ["c0", "c1"], [rows, 40], B, "uniform1", False, opt, steps=1, rtol=rtol,
                       idgen=(lambda rng, r, n: idgen(rng, r, n) if (idgen and r == rows) else rng.integers(0, r, size=n)))


@pytest.mark.parametrize("case", ["hot_multi_tile", "hot_many_tiles", "zipf_mid_table", "narrow_dense", "hot_two_in_bucket"])
def test_backward_plan_one_workgroup_heavy(dev, case):
    """tzr_tune("bwd_one_wg_heavy"): heavy buckets by one workgroup each (the setting for plans built on another stream)"""
    from torcheasyrec_amd import _lib
    assert _lib.lib().tzr_tune(b"bwd_one_wg_heavy", 1) == 0
    try:
        _plan_shape_case(dev, case)
    finally:
        _lib.lib().tzr_tune(b"bwd_one_wg_heavy", 0)


def test_backward_plan_shared_jagged_hot(dev, bwd_path):
    """two keys share a bucketed table, jagged bags, a hot row: table-major regrouping + bag_of + heavy"""
    opt = SparseOptimizerConfig(kind="rowwise_adagrad", lr=0.02)
    spec = [("u_emb", 50000, 8, "sum", ["user", "user_hist"]), ("i_emb", 3000, 16, "mean", ["item"])]
    gen = _hot(0.5, [4242])
    _run_backward_case(dev, spec, ["item", "user", "user_hist"], [3000, 50000, 50000], 300, "jagged", True, opt,
                       steps=1, rtol=5e-4, idgen=gen)


@pytest.mark.parametrize("case", ["last_row_95", "two_hot_rows", "candidate_not_hot", "shared_table", "rowwise", "sgd_small_ch"])
def test_direct_backward_hot_row_is_shared_by_the_tables_workgroups(dev, case):
    """tzr_pooled_bwd_direct: a row with more lookups than an LDS unit (the shared row of a zero-collision hash's unseen ids: the
    LAST row of the table) is summed by ALL of the table's workgroups, a slice of the positions each, and applied by the last to
    arrive; a second hot row that is not the sample's mode still goes the streaming way; a frequent row below a unit is left
    to its range"""
    from torcheasyrec_amd import _lib
    rows = 1 << 20
    kind, frac, hot_ids, B, ch, keys, spec = "adagrad", 0.95, [rows - 1], 3000, 0, ["c0", "c1"], None
    if case == "two_hot_rows":
        frac, hot_ids, B = 0.9, [rows - 1, 4321], 4000
    elif case == "candidate_not_hot":
        frac, B = 0.25, 2400
    elif case == "shared_table":  # two keys read the table: the slices cross the boundary between the keys' segments
        spec = [("t_a", rows, 16, "sum", ["c0", "c2"]), ("t_small", 40, 16, "sum", ["c1"])]
        keys, B = ["c0", "c1", "c2"], 1700
    elif case == "rowwise":
        kind = "rowwise_adagrad"
    elif case == "sgd_small_ch":
        kind, ch = "sgd", 64
    spec = spec or [("t_a", rows, 16, "sum", ["c0"]), ("t_small", 40, 16, "sum", ["c1"])]
    gen = _hot(frac, hot_ids)
    opt = SparseOptimizerConfig(kind=kind, lr=0.05, **({"initial_accumulator_value": 0.1} if kind == "adagrad" else {}))
    L = _lib.lib()
    # (bwd_direct_hot 2 = look for a hot row whether or not the caller said so: the collection of _run_backward_case does not)
    assert L.tzr_tune(b"bwd_direct", 1) == 0 and L.tzr_tune(b"bwd_direct_ch", ch) == 0 and L.tzr_tune(b"bwd_direct_hot", 2) == 0
    try:
        _run_backward_case(dev, spec, keys, [rows if k != "c1" else 40 for k in keys], B, "uniform1", False, opt, steps=2, rtol=5e-4,
                           idgen=(lambda rng, r, n: gen(rng, r, n) if r == rows else rng.integers(0, r, size=n)))
    finally:
        L.tzr_tune(b"bwd_direct", 0)
        L.tzr_tune(b"bwd_direct_ch", 0)
        L.tzr_tune(b"bwd_direct_hot", 1)


@pytest.mark.parametrize("rows,B", [(20, 8192), (24, 8000), (32, 8192), (100, 8192)])
def test_direct_backward_hot_row_with_more_workgroups_than_rows(dev, rows, B):
    """a table with rows <= k < 2 rows workgroups (k = ceil(lookups / 256): 20 or 24 rows at 8 192 lookups)
    has workgroups without a row of their own; they used to leave before the hot row's arrival count, which waits for all k:
    the hot row was never applied and the counter stayed non-zero in the reused workspace.  They now walk the ids with an
    empty range and take their slice.  Two steps: the second would read a stale counter."""
    from torcheasyrec_amd import _lib
    spec = [("t_a", rows, 16, "sum", ["c0"]), ("t_b", 5000, 16, "sum", ["c1"])]
    gen = _hot(0.6, [rows // 3])
    opt = SparseOptimizerConfig(kind="adagrad", lr=0.05, initial_accumulator_value=0.1)
    L = _lib.lib()
    assert L.tzr_tune(b"bwd_direct", 1) == 0 and L.tzr_tune(b"bwd_direct_hot", 2) == 0
    try:
        _run_backward_case(dev, spec, ["c0", "c1"], [rows, 5000], B, "uniform1", False, opt, steps=2, rtol=5e-4,
                           idgen=(lambda rng, r, n: gen(rng, r, n) if r == rows else rng.integers(0, r, size=n)))
    finally:
        L.tzr_tune(b"bwd_direct", 0)
        L.tzr_tune(b"bwd_direct_hot", 1)


def test_backward_plan_prep_fallback(dev):
    """more than BWD_GEO lookups/tables take the single-workgroup geometry kernel: forced here"""
    from torcheasyrec_amd import _lib
    opt = SparseOptimizerConfig(kind="adagrad", lr=0.05)
    assert _lib.lib().tzr_tune(b"bwd_force_prep", 1) == 0
    try:
        _run_backward_case(dev, SPEC_MIXED, ["ctx", "item", "unused_key", "user", "user_hist", "wide_user"],
                           [40, 57, 10, 1000, 1000, 1000], 45, "jagged", False, opt, steps=1)
    finally:
        _lib.lib().tzr_tune(b"bwd_force_prep", 0)


def test_backward_plan_is_bit_reproducible(dev, bwd_path):
    """same ids, same gradients -> bit-identical weights, hot rows and heavy buckets included"""
    rng = np.random.default_rng(5)
    rows, B = 100000, (1500 if dev.type == "cuda" else 700)
    ids = _hot(0.5, [99, 12345])(rng, rows, B)
    kjt = KeyedJaggedTensor(["a"], torch.from_numpy(ids.astype(np.int64)), torch.ones(B, dtype=torch.int32), uniform_length=1)
    g = torch.randn(B, 16, generator=torch.Generator().manual_seed(1))
    outs = []
    for _ in range(2):
        cfgs, _ = _make_tables([("t", rows, 16, "sum", ["a"])])
        ebc = EmbeddingBagCollection(cfgs, device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.1))
        (ebc(kjt.to(dev)).values() * g.to(dev)).sum().backward()
        outs.append(ebc.table_weights()["t"].detach().cpu().clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("kind,weighted", [("adagrad", False), ("rowwise_adagrad", True)])
def test_backward_jagged_shared_table(dev, kind, weighted, bwd_path):
    opt = SparseOptimizerConfig(kind=kind, lr=0.02, gradient_clipping=True, max_gradient=0.7)
    keys = ["ctx", "item", "unused_key", "user", "user_hist", "wide_user"]
    rows = [40, 57, 10, 1000, 1000, 1000]
    _run_backward_case(dev, SPEC_MIXED, keys, rows, 45, "jagged", weighted, opt)


def test_backward_grouped_sums_group_grads(dev, bwd_path):
    spec = [
        ("a_emb", 100, 16, "sum", ["a"]),
        ("b_emb", 7, 16, "sum", ["b"]),
        ("a_emb_wide", 100, 4, "sum", ["a"]),
        ("b_emb_wide", 7, 4, "sum", ["b"]),
    ]
    groups = {
        "wide": ["a@a_emb_wide", "b@b_emb_wide"],
        "fm": ["a@a_emb", "b@b_emb"],
        "deep": ["b@b_emb", "a@a_emb"],
    }
    opt = SparseOptimizerConfig(kind="adagrad", lr=0.01)
    _run_backward_case(dev, spec, ["a", "b"], [100, 7], 64, "uniform1", False, opt, groups=groups)


def test_rowwise_weight_decay_modes(dev, bwd_path):
    for mode in ("l2", "decouple"):
        opt = SparseOptimizerConfig(kind="rowwise_adagrad", lr=0.03, weight_decay=0.01, weight_decay_mode=mode)
        _run_backward_case(dev, SPEC_CRITEO_SMALL, ["c0", "c1", "c2", "c3"], [5000, 300, 3, 4], 100,
                           "uniform1", False, opt, steps=2)


def test_frozen_table_is_left_out_of_the_fused_optimizer(dev):
    """`trainable: false` (tzrec/features/feature.py:629, models/model.py:162-201): the table is read
    in forward, never written in backward; its neighbours update exactly as if it were not there."""
    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    rng = np.random.default_rng(3)
    B = 50
    lens = rng.integers(0, 4, size=3 * B).astype(np.int32)
    rows = [40, 7, 300]
    vals = np.concatenate([rng.integers(0, rows[f], size=int(lens[f * B:(f + 1) * B].sum())) for f in range(3)])
    kjt = KeyedJaggedTensor(["a", "b", "c"], torch.from_numpy(vals.astype(np.int64)), torch.from_numpy(lens)).to(dev)
    res = {}
    for frozen in (False, True):
        torch.manual_seed(0)
        ebc = EmbeddingBagCollection(
            [EmbeddingBagConfig("ta", 8, rows[0], ["a"]), EmbeddingBagConfig("tb", 8, rows[1], ["b"], trainable=not frozen),
             EmbeddingBagConfig("tc", 8, rows[2], ["c"])], device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.1))
        before = {n: w.detach().clone() for n, w in ebc.table_weights().items()}
        g = torch.randn(B, 24, generator=torch.Generator().manual_seed(1)).to(dev)
        (ebc(kjt).values() * g).sum().backward()
        res[frozen] = ({n: w.detach().clone() for n, w in ebc.table_weights().items()}, before)
    after_f, before_f = res[True]
    assert torch.equal(after_f["tb"], before_f["tb"])  # frozen: bit-identical
    assert not torch.equal(res[False][0]["tb"], res[False][1]["tb"])  # trainable twin did move
    for n in ("ta", "tc"):
        assert torch.equal(after_f[n], res[False][0][n])  # neighbours: same update either way


@pytest.mark.parametrize("kind", ["sgd", "adagrad", "rowwise_adagrad"])
def test_fp16_tables(dev, kind, bwd_path):
    """`data_type: FP16` (tzrec/features/feature.py:346-356): half rows are widened exactly on read (so
    L=1 pooling is still a bit-exact copy), the optimizer computes in fp32 and rounds to nearest even on
    the way back; an fp32 table in the same collection is untouched by all of that."""
    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    rng = np.random.default_rng(11)
    B, D, lr = 200, 16, 0.1
    rows = [50, 7]
    ebc = EmbeddingBagCollection(
        [EmbeddingBagConfig("half_t", D, rows[0], ["a"], data_type="FP16"), EmbeddingBagConfig("float_t", D, rows[1], ["b"])],
        device=dev, optimizer=SparseOptimizerConfig(kind=kind, lr=lr))
    wh = ebc.table_weights()["half_t"]
    assert wh.dtype == torch.float16
    ids = np.stack([rng.integers(0, rows[0], size=B), rng.integers(0, rows[1], size=B)]).astype(np.int64)
    kjt = KeyedJaggedTensor(["a", "b"], torch.from_numpy(ids.reshape(-1)), torch.ones(2 * B, dtype=torch.int32), uniform_length=1).to(dev)
    w0 = [wh.detach().cpu().float().numpy().copy(), ebc.table_weights()["float_t"].detach().cpu().numpy().copy()]
    out = ebc(kjt).values()
    assert torch.equal(out.detach().cpu()[:, :D], torch.from_numpy(w0[0])[ids[0]])  # exact widening
    assert torch.equal(out.detach().cpu()[:, D:], torch.from_numpy(w0[1])[ids[1]])
    g = torch.randn(B, 2 * D, generator=torch.Generator().manual_seed(2))
    (out * g.to(dev)).sum().backward()
    opt = orc.SparseOptim(kind=kind, lr=lr)
    for t, name in enumerate(["half_t", "float_t"]):
        w = w0[t].copy()
        m = None if kind == "sgd" else (np.zeros_like(w) if kind == "adagrad" else np.zeros(rows[t], np.float32))
        orc.sparse_update(w, m, ids[t], g[:, t * D:(t + 1) * D].numpy(), opt)
        got = ebc.table_weights()[name].detach().cpu()
        if t == 0:
            assert got.dtype == torch.float16
            want = torch.from_numpy(w).half()
            # one fp32 ulp in the update can flip the half rounding of an element: allow 1 half ulp
            torch.testing.assert_close(got.float(), want.float(), rtol=1e-3, atol=1e-5)
            exact = (got == want).float().mean().item()
            assert exact > 0.98, exact
        else:
            np.testing.assert_allclose(got.numpy(), w, rtol=2e-5, atol=2e-3 * lr)
        if m is not None:
            np.testing.assert_allclose(ebc.table_states()[name].detach().cpu().numpy(), m, rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("kind,B", [("adagrad", 200), ("rowwise_adagrad", 1500), ("sgd", 33)])
def test_forward_launch_that_carries_the_plan(dev, kind, B, monkeypatch):
    """tzr_pooled_fwd_cells_plan: the one-id forward and the cells plan of the same batch as ONE launch (forward and partition
    workgroups alternate in the grid) -- against the oracle like every other form, and bit for bit against the two launches."""
    from torcheasyrec_amd import _lib

    L = _lib.lib()
    monkeypatch.setenv("TZR_BWD_PLAN", "cells")
    assert L.tzr_tune(b"bwd_direct", -1) == 0 and L.tzr_tune(b"fwd_plan", 2) == 0  # (at test sizes: no one-launch backward, any batch size)
    try:
        opt = SparseOptimizerConfig(kind=kind, lr=0.05)
        keys, rows = ["c0", "c1", "c2", "c3"], [5000, 300, 3, 4]
        made = []
        real = EmbeddingBagCollection.__init__

        def spy(self, *a, **k):
            real(self, *a, **k)
            made.append(self)

        monkeypatch.setattr(EmbeddingBagCollection, "__init__", spy)
        _run_backward_case(dev, SPEC_CRITEO_SMALL, keys, rows, B, "uniform1", False, opt)
        assert made and made[0].forward_plans == 2  # both steps
        monkeypatch.setattr(EmbeddingBagCollection, "__init__", real)
        # the same two steps with and without: same bits everywhere
        res = []
        for carry in (True, False):
            cfgs, _ = _make_tables(SPEC_CRITEO_SMALL)
            ebc = EmbeddingBagCollection(cfgs, device=dev, optimizer=opt)
            ebc.forward_plan = carry
            rng = np.random.default_rng(11)
            outs = []
            for _ in range(2):
                kd = _make_kjt(keys, rows, B, rng).to(dev)
                out = ebc(kd).values()
                g = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).to(dev)
                (out * g).sum().backward()
                outs.append(out.detach().cpu())
            assert ebc.forward_plans == (2 if carry else 0)
            res.append((outs, {n: w.detach().cpu() for n, w in ebc.table_weights().items()},
                        {n: (s.detach().cpu() if s is not None else None) for n, s in ebc.table_states().items()}))
        for a, b in zip(res[0][0], res[1][0]):
            assert torch.equal(a, b)
        for n in res[0][1]:
            assert torch.equal(res[0][1][n], res[1][1][n]), n
            if res[0][2].get(n) is not None:
                assert torch.equal(res[0][2][n], res[1][2][n]), n
        # a forward whose backward never ran leaves a plan behind; the ids are then refreshed IN PLACE (a static input buffer): the
        # next forward of the same object plans again instead of handing the stale plan to its backward
        cfgs, _ = _make_tables(SPEC_CRITEO_SMALL)
        e1 = EmbeddingBagCollection(cfgs, device=dev, optimizer=opt)
        e2 = EmbeddingBagCollection(cfgs, device=dev, optimizer=opt)
        rng = np.random.default_rng(5)
        k1 = _make_kjt(keys, rows, B, rng).to(dev)
        fresh = _make_kjt(keys, rows, B, rng).to(dev)
        e1(k1)  # (plan made, never consumed)
        assert k1._tzr_plan is not None
        k1.values().copy_(fresh.values())
        g = torch.from_numpy(rng.standard_normal((B, 64)).astype(np.float32)).to(dev)
        (e1(k1).values() * g).sum().backward()
        (e2(fresh).values() * g).sum().backward()
        assert e1.forward_plans == 2
        for n in e1.table_weights():
            assert torch.equal(e1.table_weights()[n].cpu(), e2.table_weights()[n].cpu()), n
        # an evaluation forward (no backward follows) stays the plain launch
        ebc.forward_plan = True
        with torch.no_grad():
            ebc(kd)
        assert ebc.forward_plans == 0
    finally:
        L.tzr_tune(b"bwd_direct", 0)
        L.tzr_tune(b"fwd_plan", 1)


def test_forward_that_carries_the_plan_with_a_frozen_table_and_what_it_declines(dev, monkeypatch):
    """A frozen table: the forward reads it (forward descriptors), the plan in the same launch leaves it out (backward descriptors) --
    same bits as the two launches.  Per-sample weights, half-precision tables, jagged bags: the forward goes out on its own."""
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig
    from torcheasyrec_amd.sparse import KeyedJaggedTensor

    L = _lib.lib()
    monkeypatch.setenv("TZR_BWD_PLAN", "cells")
    assert L.tzr_tune(b"bwd_direct", -1) == 0 and L.tzr_tune(b"fwd_plan", 2) == 0
    try:
        rng = np.random.default_rng(3)
        B, rows = 300, [40, 7, 3000]
        ids = np.stack([rng.integers(0, r, size=B) for r in rows]).astype(np.int64)
        kjt = KeyedJaggedTensor(["a", "b", "c"], torch.from_numpy(ids.reshape(-1)), torch.ones(3 * B, dtype=torch.int32)).to(dev)
        g = torch.randn(B, 24, generator=torch.Generator().manual_seed(1)).to(dev)
        res = []
        for carry in (True, False):
            torch.manual_seed(0)
            ebc = EmbeddingBagCollection(
                [EmbeddingBagConfig("ta", 8, rows[0], ["a"]), EmbeddingBagConfig("tb", 8, rows[1], ["b"], trainable=False),
                 EmbeddingBagConfig("tc", 8, rows[2], ["c"])], device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.1))
            ebc.forward_plan = carry
            before_tb = ebc.table_weights()["tb"].detach().clone()
            out = ebc(kjt).values()
            (out * g).sum().backward()
            assert ebc.forward_plans == (1 if carry else 0)
            assert torch.equal(ebc.table_weights()["tb"].detach(), before_tb)  # frozen: never written
            res.append((out.detach().cpu(), {n: w.detach().cpu().clone() for n, w in ebc.table_weights().items()}))
        assert torch.equal(res[0][0], res[1][0])
        for n in res[0][1]:
            assert torch.equal(res[0][1][n], res[1][1][n]), n
        # what the launch does not take
        opt = SparseOptimizerConfig(kind="adagrad", lr=0.1)
        half = EmbeddingBagCollection([EmbeddingBagConfig("h", 8, 40, ["a"], data_type="FP16")], device=dev, optimizer=opt)
        ka = KeyedJaggedTensor(["a"], torch.from_numpy(ids[0]), torch.ones(B, dtype=torch.int32)).to(dev)
        half(ka).values().sum().backward()
        assert half.forward_plans == 0
        plain = EmbeddingBagCollection([EmbeddingBagConfig("p", 8, 40, ["a"])], device=dev, optimizer=opt)
        kw = KeyedJaggedTensor(["a"], torch.from_numpy(ids[0]), torch.ones(B, dtype=torch.int32), weights=torch.rand(B)).to(dev)
        plain(kw).values().sum().backward()
        lens = rng.integers(0, 3, size=B).astype(np.int32)
        kj = KeyedJaggedTensor(["a"], torch.from_numpy(rng.integers(0, 40, size=int(lens.sum())).astype(np.int64)), torch.from_numpy(lens)).to(dev)
        plain(kj).values().sum().backward()
        assert plain.forward_plans == 0
        plain(ka).values().sum().backward()
        assert plain.forward_plans == 1
    finally:
        L.tzr_tune(b"bwd_direct", 0)
        L.tzr_tune(b"fwd_plan", 1)
