"""Index stage (K1-K4): bit-exact against the oracle, incl. empty / ragged / boundary cases."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor, block_bucketize, lengths_to_offsets  # noqa: E402


@pytest.mark.parametrize("n", [0, 1, 63, 2048, 2049, 70000])
@pytest.mark.parametrize("dtype", [torch.int32, torch.int64])
def test_lengths_to_offsets(dev, n, dtype):
    if n == 70000 and dev.type == "cpu" and dtype == torch.int64:
        pytest.skip("covered by int32 on the emulator")
    rng = np.random.default_rng(n)
    lengths = torch.from_numpy(rng.integers(0, 9, size=n)).to(dtype)
    got = lengths_to_offsets(lengths.to(dev)).cpu().numpy()
    ref = orc.lengths_to_offsets(lengths.numpy())
    assert got.dtype == np.int64 and np.array_equal(got, ref)


def _rand_kjt(rng, F, B, max_len, rows, weighted):
    lens = rng.integers(0, max_len + 1, size=F * B).astype(np.int32)
    if B > 3:
        lens[rng.integers(0, F * B, size=F)] = 0
    vals = rng.integers(0, rows, size=int(lens.sum())).astype(np.int64)
    w = rng.uniform(0, 1, size=len(vals)).astype(np.float32) if weighted else None
    keys = [f"k{i}" for i in range(F)]
    return keys, vals, lens, w


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("perm", [[2, 0, 1, 3], [3, 3, 0], [1], [0, 1, 2, 3]])
def test_kjt_permute(dev, perm, weighted):
    rng = np.random.default_rng(len(perm))
    F, B = 4, 33
    keys, vals, lens, w = _rand_kjt(rng, F, B, 5, 1000, weighted)
    kjt = KeyedJaggedTensor(keys, torch.from_numpy(vals), torch.from_numpy(lens),
                            torch.from_numpy(w) if weighted else None).to(dev)
    out = kjt.permute(perm)
    rl, rv, rw = orc.kjt_permute(perm, lens, vals, w, B)
    assert out.keys() == [keys[i] for i in perm]
    assert np.array_equal(out.lengths().cpu().numpy(), rl)
    assert np.array_equal(out.values().cpu().numpy(), rv)
    assert np.array_equal(out.offsets().cpu().numpy(), orc.lengths_to_offsets(rl))
    if weighted:
        assert np.array_equal(out.weights().cpu().numpy(), rw)


@pytest.mark.parametrize("W", [1, 2, 8])
@pytest.mark.parametrize("weighted", [False, True])
def test_block_bucketize(dev, W, weighted):
    rng = np.random.default_rng(W)
    F, B = 3, 41
    rows = [1000, 17, 40_000_000]
    keys = [f"k{i}" for i in range(F)]
    lens = rng.integers(0, 5, size=F * B).astype(np.int32)
    off = orc.lengths_to_offsets(lens)
    vals = np.concatenate([rng.integers(0, rows[f], size=int(off[(f + 1) * B] - off[f * B])) for f in range(F)]).astype(np.int64)
    w = rng.uniform(0, 1, size=len(vals)).astype(np.float32) if weighted else None
    block = np.array([(r + W - 1) // W for r in rows], dtype=np.int64)  # ceil(rows / W)
    kjt = KeyedJaggedTensor(keys, torch.from_numpy(vals), torch.from_numpy(lens),
                            torch.from_numpy(w) if weighted else None).to(dev)
    out, unb = block_bucketize(kjt, torch.from_numpy(block).to(dev), W, return_permute=True)
    rl, rv, rw, ru = orc.block_bucketize(block, lens, vals, w, B, W)
    assert np.array_equal(out.lengths().cpu().numpy(), rl)
    assert np.array_equal(out.values().cpu().numpy(), rv)
    assert np.array_equal(unb.cpu().numpy(), ru)
    if weighted:
        assert np.array_equal(out.weights().cpu().numpy(), rw)
    # owner rank r only ever sees local ids inside its block
    no = orc.lengths_to_offsets(rl)
    for r in range(W):
        for f in range(F):
            seg = rv[no[(r * F + f) * B]: no[(r * F + f + 1) * B]]
            assert (seg >= 0).all() and (seg < block[f]).all()


@pytest.mark.parametrize("mode", [_lib.BOUNDS_FATAL, _lib.BOUNDS_WARNING, _lib.BOUNDS_IGNORE])
def test_bounds_check(dev, mode):
    rng = np.random.default_rng(7)
    B = 50
    cfgs = [EmbeddingBagConfig("t0", 16, 10, ["a"]), EmbeddingBagConfig("t1", 16, 1000, ["b"])]
    ebc = EmbeddingBagCollection(cfgs, device=dev)
    lens = rng.integers(0, 4, size=2 * B).astype(np.int32)
    off = orc.lengths_to_offsets(lens)
    vals = np.concatenate([rng.integers(-2, 14, size=int(off[B])), rng.integers(990, 1010, size=int(off[2 * B] - off[B]))]).astype(np.int64)
    kjt = KeyedJaggedTensor(["a", "b"], torch.from_numpy(vals.copy()), torch.from_numpy(lens)).to(dev)
    cnt = ebc.bounds_check(kjt, mode)
    ref_vals, ref_bad = orc.bounds_check(vals, off, [10, 1000], B, clamp=mode != _lib.BOUNDS_FATAL)
    assert int(cnt.item()) == (0 if mode == _lib.BOUNDS_IGNORE else ref_bad)
    assert np.array_equal(kjt.values().cpu().numpy(), ref_vals)


@pytest.mark.parametrize("W", [1, 2, 8])
def test_block_bucketize_hash_routing(dev, W):
    """block_size 0 = hash routing of raw 64-bit ids (ZCH tables): owner = splitmix64(id) mod W, the id
    travels unchanged; other keys of the same call keep block routing."""
    rng = np.random.default_rng(W + 50)
    F, B = 2, 57
    lens = rng.integers(0, 4, size=F * B).astype(np.int32)
    off = orc.lengths_to_offsets(lens)
    raw = rng.integers(-(1 << 62), 1 << 62, size=int(off[B])).astype(np.int64)  # any int64, negatives too
    raw[:3] = [0, -1, (1 << 63) - 2]
    blocked = rng.integers(0, 1000, size=int(off[2 * B] - off[B])).astype(np.int64)
    vals = np.concatenate([raw, blocked])
    block = np.array([0, (1000 + W - 1) // W], dtype=np.int64)
    kjt = KeyedJaggedTensor(["zch_key", "plain"], torch.from_numpy(vals), torch.from_numpy(lens)).to(dev)
    out, unb = block_bucketize(kjt, torch.from_numpy(block).to(dev), W, return_permute=True)
    rl, rv, _, ru = orc.block_bucketize(block, lens, vals, None, B, W)
    assert np.array_equal(out.lengths().cpu().numpy(), rl)
    assert np.array_equal(out.values().cpu().numpy(), rv)
    assert np.array_equal(unb.cpu().numpy(), ru)
    # every raw id arrives unchanged at exactly the rank its hash names
    no = orc.lengths_to_offsets(rl)
    got = []
    for r in range(W):
        seg = rv[no[(r * F) * B]: no[(r * F + 1) * B]]
        assert ((orc.splitmix64(seg) % np.uint64(W)).astype(np.int64) == r).all()
        got.append(seg)
    assert sorted(np.concatenate(got).tolist()) == sorted(raw.tolist())
    if W == 8:  # the hash spreads: no rank is starved or flooded
        counts = np.array([len(g) for g in got])
        assert counts.min() > 0


@pytest.mark.parametrize("W,hashed", [(1, False), (8, False), (5, True)])
def test_exchange_bucketize_equals_permute_then_bucketize(dev, W, hashed):
    """The lean 3-launch bucketize of the sharded exchange (uniform bags, selected keys) gives exactly
    what K1 permute + K2 bucketize give: ids by (rank, key) in lookup order, positions, counts."""
    rng = np.random.default_rng(W * 3 + hashed)
    F, B = 5, 1100  # 2 tiles per key, ragged last tile
    rows = [1000, 17, 40_000_000, 300, 9]
    keys = [f"k{i}" for i in range(F)]
    vals = np.stack([rng.integers(0, rows[f], size=B) for f in range(F)]).astype(np.int64)
    if hashed:
        vals[1] = rng.integers(-(1 << 62), 1 << 62, size=B)
    kjt = KeyedJaggedTensor(keys, torch.from_numpy(vals.reshape(-1).copy()), torch.ones(F * B, dtype=torch.int32), uniform_length=1).to(dev)
    sel = [3, 1, 2]  # a subset, reordered
    block = np.array([(rows[f] + W - 1) // W for f in sel], dtype=np.int64)
    if hashed:
        block[1] = 0  # key k1 routed by hash
    rot = np.array([1 % W, 0, (W - 1)], dtype=np.int32)
    d_blk, d_rot = torch.from_numpy(block).to(dev), torch.from_numpy(rot).to(dev)
    want, want_unb = block_bucketize(kjt.permute(sel), d_blk, W, return_permute=True, rank_offsets=d_rot)
    L = _lib.lib()
    n = len(sel) * B
    out = torch.empty(n, dtype=torch.int64, device=dev)
    unb = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.empty(W * len(sel), dtype=torch.int64, device=dev)
    ws = _lib.workspace(L.tzr_exchange_bucketize_workspace(len(sel), B, W), dev)
    d_sel = torch.tensor(sel, dtype=torch.int32, device=dev)
    _lib.check(L.tzr_exchange_bucketize(_lib.ptr(d_sel), len(sel), _lib.ptr(d_blk), _lib.ptr(d_rot), B, 1, W, _lib.ptr(kjt.values()),
                                        _lib.ptr(out), _lib.ptr(unb), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
               "tzr_exchange_bucketize")
    assert torch.equal(out, want.values())
    assert torch.equal(unb, want_unb)
    off = want.offsets()
    assert torch.equal(cnt, off[B::B] - off[:-1:B])


def _capped_message_from_dense(ids, unb, cnt, W, F, cap):
    """numpy restatement of the capacity-bounded layout (include/tzrec_hip.h: tzr_exchange_bucketize_capped) from the
    dense bucketize result: per destination the clamped counts, the overflow word, the first `cap` ids."""
    S = F + 1 + cap
    msg = np.full(W * S, -7, dtype=np.int64)  # -7: words the kernel must leave alone
    cnt = cnt.reshape(W, F)
    start = np.concatenate([[0], np.cumsum(cnt.reshape(-1))])
    over = int((cnt.sum(1) > cap).any())
    new_pos = np.empty(len(ids), dtype=np.int64)
    for d in range(W):
        run = 0
        for f in range(F):
            keep = min(int(cnt[d, f]), cap - run)
            msg[d * S + f] = keep
            s0 = start[d * F + f]
            msg[d * S + F + 1 + run: d * S + F + 1 + run + keep] = ids[s0:s0 + keep]
            new_pos[s0:s0 + keep] = d * S + F + 1 + run + np.arange(keep)
            new_pos[s0 + keep:s0 + int(cnt[d, f])] = d * S + F  # dropped
            run += keep
        msg[d * S + F] = over
    return msg, new_pos[unb], over


@pytest.mark.parametrize("W,cap,hashed", [(1, 4000, False), (4, 2000, False), (4, 700, False), (8, 100, True), (3, 1, False)])
def test_exchange_bucketize_capped(dev, W, cap, hashed):
    """Fixed-capacity message layout = the dense bucketize re-laid per destination; overflow clamps and flags;
    the owner-side segments kernel turns the received headers into key segments with dead gaps."""
    rng = np.random.default_rng(W * 5 + cap)
    F, B = 3, 1100
    rows = [1000, 17, 40_000_000, 300, 9]
    keys = [f"k{i}" for i in range(5)]
    vals = np.stack([rng.integers(0, rows[f], size=B) for f in range(5)]).astype(np.int64)
    if hashed:
        vals[1] = rng.integers(-(1 << 62), 1 << 62, size=B)
    kjt = KeyedJaggedTensor(keys, torch.from_numpy(vals.reshape(-1).copy()), torch.ones(5 * B, dtype=torch.int32), uniform_length=1).to(dev)
    sel = [3, 1, 2]
    block = np.array([(rows[f] + W - 1) // W for f in sel], dtype=np.int64)
    if hashed:
        block[1] = 0
    rot = np.array([1 % W, 0, (W - 1)], dtype=np.int32)
    d_blk, d_rot = torch.from_numpy(block).to(dev), torch.from_numpy(rot).to(dev)
    L = _lib.lib()
    n = F * B
    out = torch.empty(n, dtype=torch.int64, device=dev)
    unb = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.empty(W * F, dtype=torch.int64, device=dev)
    ws = _lib.workspace(L.tzr_exchange_bucketize_workspace(F, B, W), dev)
    d_sel = torch.tensor(sel, dtype=torch.int32, device=dev)
    _lib.check(L.tzr_exchange_bucketize(_lib.ptr(d_sel), F, _lib.ptr(d_blk), _lib.ptr(d_rot), B, 1, W, _lib.ptr(kjt.values()),
                                        _lib.ptr(out), _lib.ptr(unb), _lib.ptr(cnt), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
               "tzr_exchange_bucketize")
    S = L.tzr_exchange_message_stride(F, cap)
    assert S == F + 1 + cap
    want_msg, want_unb, want_over = _capped_message_from_dense(out.cpu().numpy(), unb.cpu().numpy(), cnt.cpu().numpy(), W, F, cap)
    msg = torch.full((W * S,), -7, dtype=torch.int64, device=dev)
    unb2 = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(L.tzr_exchange_bucketize_capped(_lib.ptr(d_sel), F, _lib.ptr(d_blk), _lib.ptr(d_rot), B, 1, W, _lib.ptr(kjt.values()),
                                               cap, _lib.ptr(msg), _lib.ptr(unb2), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)),
               "tzr_exchange_bucketize_capped")
    assert np.array_equal(msg.cpu().numpy(), want_msg)
    assert np.array_equal(unb2.cpu().numpy(), want_unb)
    assert want_over == int(cap in (700, 100, 1))  # the cases meant to overflow do
    # the same layout from the dense result (the path ragged / weighted bags take): tzr_exchange_pad
    # (round 2 kept this call out of the `-m gpu` selection after one unexplained process death next to
    # tests/test_sharded_gpu.py; the graph captures there are thread_local now, NOTES.md)
    msg3 = torch.full((W * S,), -7, dtype=torch.int64, device=dev)
    unb3 = torch.empty(n, dtype=torch.int64, device=dev)
    _lib.check(L.tzr_exchange_pad(_lib.ptr(cnt), W, F, cap, _lib.ptr(out), _lib.ptr(unb), n, _lib.ptr(msg3), _lib.ptr(unb3),
                                  _lib.stream_ptr(dev)), "tzr_exchange_pad")
    assert torch.equal(msg3, msg) and torch.equal(unb3, unb2)
    # owner side (a world where every rank sent this very message): key segments + the flag
    ks = torch.full((W * (F + 1) + 2,), -1, dtype=torch.int64, device=dev)
    flag = torch.full((1,), -1, dtype=torch.int64, device=dev)
    _lib.check(L.tzr_exchange_owner_segments(_lib.ptr(msg), W, F, cap, _lib.ptr(ks), _lib.ptr(flag), _lib.stream_ptr(dev)),
               "tzr_exchange_owner_segments")
    ks, m = ks.cpu().numpy(), want_msg
    assert int(flag.item()) == want_over
    assert ks[0] == 0 and ks[-1] == W * S and np.all(np.diff(ks) >= 0)
    for s in range(W):
        run = s * S + F + 1
        for f in range(F):
            assert ks[s * (F + 1) + 1 + f] == run
            run += m[s * S + f]
        assert ks[(s + 1) * (F + 1)] == run  # the next dead key starts where this rank's ids end


@pytest.mark.parametrize("n_keys,per_key,dim", [(7, 500, 16), (40, 33, 8), (300, 9, 16), (3, 1, 4), (600, 2, 12), (1500, 0, 16)])
def test_rows_gather_matches_indexing(dev, n_keys, per_key, dim):
    """tzr_rows_gather (owner side of the id-granularity exchange): out[j] = W_table(key(j))[ids[j]] over key segments,
    several workgroups and several keys per workgroup (1 500 keys, half of them empty: more segments in one workgroup's rows
    than its LDS stages), fp32 and fp16 tables of different widths, dead keys left untouched, out-of-range ids read row 0, a padded
    output stride."""
    rng = np.random.default_rng(n_keys * 31 + dim)
    T = 4
    rows = [11, 257, 1000, 5]
    dims = [dim, dim, max(4, dim - 4), dim]
    ws = [torch.from_numpy(rng.standard_normal((rows[t], dims[t])).astype(np.float32)) for t in range(T)]
    ws[1] = ws[1].half()
    wd = [w.to(dev) for w in ws]
    tables = np.zeros(T, dtype=_lib.TABLE_DT)
    for t in range(T):
        tables[t]["w"], tables[t]["rows"], tables[t]["dim"] = wd[t].data_ptr(), rows[t], dims[t]
        tables[t]["w_stride"], tables[t]["w_dtype"] = wd[t].stride(0), _lib.DT_F16 if wd[t].dtype == torch.float16 else _lib.DT_F32
    key_table = rng.integers(-1, T, size=n_keys).astype(np.int32)
    counts = rng.integers(0, 2 * per_key + 1, size=n_keys) if per_key else rng.integers(0, 2, size=n_keys)
    key_start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    n = int(key_start[-1])
    ids = np.zeros(n, dtype=np.int64)
    for k in range(n_keys):
        t = key_table[k]
        hi = rows[t] if t >= 0 else 10
        ids[key_start[k]:key_start[k + 1]] = rng.integers(0, hi, size=counts[k])
    if n:
        ids[rng.integers(0, n, size=max(1, n // 50))] = 10 ** 9  # out of range: row 0, as the pooled forward does
    stride = dim + 4
    out = torch.full((max(n, 1), stride), -7.0, dtype=torch.float32, device=dev)
    # (named: a temporary's block would go back to the caching allocator -- and to the next upload -- before the launch)
    d_tables, d_kt, d_ks, d_ids = (_lib.upload_struct(tables, dev), torch.from_numpy(key_table).to(dev), torch.from_numpy(key_start).to(dev),
                                   torch.from_numpy(ids).to(dev))
    rc = _lib.lib().tzr_rows_gather(_lib.ptr(d_tables), _lib.ptr(d_kt), _lib.ptr(d_ks), n_keys, _lib.ptr(d_ids), n,
                                    _lib.ptr(out), stride, dim, _lib.stream_ptr(dev))
    _lib.check(rc, "tzr_rows_gather")
    if dev.type == "cuda":
        torch.cuda.synchronize()
    want = np.full((max(n, 1), stride), -7.0, dtype=np.float32)
    for k in range(n_keys):
        t = key_table[k]
        if t < 0:
            continue
        for j in range(key_start[k], key_start[k + 1]):
            i = ids[j] if 0 <= ids[j] < rows[t] else 0
            want[j, :dim] = 0.0
            want[j, :dims[t]] = ws[t][i].float().numpy()
    assert np.array_equal(out.cpu().numpy(), want)
