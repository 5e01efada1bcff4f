"""Sequence path: unpooled EmbeddingCollection lookup, jagged->padded dense,
DIN target attention, and the fused sparse update through it -- against the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd.embedding import SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sequence import DINEncoder, EmbeddingCollection, EmbeddingConfig, jagged_to_padded_dense  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402


def _seq_kjt(rng, keys, rows, B, max_l):
    lens = rng.integers(0, max_l + 1, size=len(keys) * B).astype(np.int32)
    lens[rng.integers(0, len(lens), size=3)] = 0
    vals = np.concatenate([rng.integers(0, rows[k], size=int(lens[i * B:(i + 1) * B].sum())) for i, k in enumerate(keys)]).astype(np.int64)
    return KeyedJaggedTensor(keys, torch.from_numpy(vals), torch.from_numpy(lens))


@pytest.mark.parametrize("max_len", [1, 5, 12])
def test_jagged_to_padded_dense(dev, max_len):
    rng = np.random.default_rng(max_len)
    B, D = 19, 8
    lens = torch.from_numpy(rng.integers(0, 9, size=B).astype(np.int64))
    lens[3] = 0
    N = int(lens.sum())
    v = torch.randn(N, D)
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(lens, 0)
    vd = v.clone().to(dev).requires_grad_(True)
    out = jagged_to_padded_dense(vd, off.to(dev), max_len, -1.5)
    vr = v.clone().requires_grad_(True)
    ref = orc.jagged_to_padded_dense(vr, lens, max_len, -1.5)
    assert torch.equal(out.detach().cpu(), ref.detach())  # a copy: bit-exact
    g = torch.randn(B, max_len, D)
    out.backward(g.to(dev))
    ref.backward(g)
    assert torch.equal(vd.grad.cpu(), vr.grad)


def test_unpooled_lookup_and_din_training_step(dev):
    """click_seq ids + target item id share one table (tzrec multi_tower_din style): lookup, pad,
    DIN attention, loss; the shared table must receive the exact fused update."""
    rng = np.random.default_rng(4)
    B, D, L, rows, lr = 23, 16, 10, 300, 0.05
    g = torch.Generator().manual_seed(1)
    w0 = (torch.rand(rows, D, generator=g) - 0.5) * 0.4
    ec = EmbeddingCollection([EmbeddingConfig("item_emb", D, rows, ["item_id", "click_seq__item_id"],
                                              init_fn=lambda t: t.copy_(w0))],
                             device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=lr))
    torch.manual_seed(5)
    din = DINEncoder(D, D, "seq", {"hidden_units": [32, 8]}).to(dev)
    lens = rng.integers(0, L + 3, size=B).astype(np.int32)  # some longer than L (truncated)
    lens[2] = 0
    seq_ids = rng.integers(0, rows, size=int(lens.sum())).astype(np.int64)
    item_ids = rng.integers(0, rows, size=B).astype(np.int64)
    kjt = KeyedJaggedTensor(["item_id", "click_seq__item_id"], torch.from_numpy(np.concatenate([item_ids, seq_ids])),
                            torch.from_numpy(np.concatenate([np.ones(B, np.int32), lens])))
    jts = ec(kjt.to(dev))
    assert sorted(jts) == ["click_seq__item_id", "item_id"]
    q = jts["item_id"].values()
    seq = jts["click_seq__item_id"].to_padded_dense(L)
    out = din({"seq.query": q, "seq.sequence": seq, "seq.sequence_length": torch.from_numpy(lens.astype(np.int64)).to(dev)})
    tgt = torch.randn(B, D, generator=g)
    loss = ((out - tgt.to(dev)) ** 2).mean()
    loss.backward()

    # oracle
    wr = w0.clone().requires_grad_(True)
    rows_q = wr[torch.from_numpy(item_ids)]
    rows_s = wr[torch.from_numpy(seq_ids)]
    torch.testing.assert_close(q.detach().cpu(), rows_q.detach(), rtol=0, atol=0)
    lin = [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in din.mlp.mlp if hasattr(m, "weight")]
    ref = orc.din_encoder(rows_q, orc.jagged_to_padded_dense(rows_s, torch.from_numpy(lens.astype(np.int64)), L),
                          torch.from_numpy(lens.astype(np.int64)), lin, (din.linear.weight.detach().cpu(), din.linear.bias.detach().cpu()))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    ref_loss = ((ref - tgt) ** 2).mean()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item()) + 1e-7
    # per-id gradients (query ids then sequence ids = KJT value order) -> exact fused update
    gq, gs = torch.autograd.grad(ref_loss, [rows_q, rows_s])
    w = w0.numpy().copy()
    m = np.zeros_like(w)
    orc.sparse_update(w, m, np.concatenate([item_ids, seq_ids]), np.concatenate([gq.numpy(), gs.numpy()], axis=0),
                      orc.SparseOptim(kind="adagrad", lr=lr))
    got = ec.table_weights()["item_emb"].detach().cpu().numpy()
    np.testing.assert_allclose(got, w, rtol=2e-4, atol=2e-3 * lr)


@pytest.mark.parametrize("pooling", ["sum", "mean"])
@pytest.mark.parametrize("D", [4, 16, 36])
def test_segment_reduce_matches_torch(dev, pooling, D):
    """multi-valued sequence steps (tzrec/modules/embedding.py:1353-1366): against
    torch.segment_reduce + nan_to_num, forward and backward, empty segments included"""
    from torcheasyrec_amd.sequence import segment_reduce

    g = torch.Generator().manual_seed(D)
    lengths = torch.tensor([2, 0, 1, 5, 0, 3, 1, 0], dtype=torch.int64)
    N = int(lengths.sum())
    x = torch.randn(N, D, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = torch.nan_to_num(torch.segment_reduce(xr, pooling, lengths=lengths), nan=0.0)
    xd = x.clone().to(dev).requires_grad_(True)
    out = segment_reduce(xd, lengths.to(dev), pooling)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    out.backward(go.to(dev))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-6, atol=1e-6)
    # no segments at all
    e = segment_reduce(torch.zeros(0, D).to(dev), torch.zeros(0, dtype=torch.int64).to(dev), pooling)
    assert tuple(e.shape) == (0, D)


def test_din_first_layer_split_equals_the_literal_input():
    """DINEncoder: W [q, k, q - k, q * k] evaluated as (Wa + Wc) q + (Wb - Wc) k + Wd (q * k) -- the query part once per sample,
    half the contraction per position, no [B, L, 4 D] tensor -- against the reference's literal concatenation
    (/root/reference/tzrec/modules/sequence.py:100-128): output and every gradient to fp32 rounding."""
    import torch

    from torcheasyrec_amd.sequence import DINEncoder

    torch.manual_seed(0)
    for qd in (32, 16):
        enc = DINEncoder(32, qd, "seq", {"hidden_units": [48, 8]})
        B, L = 7, 11
        emb = {"seq.query": torch.randn(B, qd, requires_grad=True), "seq.sequence": torch.randn(B, L, 32, requires_grad=True),
               "seq.sequence_length": torch.tensor([0, 3, 11, 5, 1, 9, 11])}
        outs = {}
        for split in (True, False):
            enc.split_first_layer = split
            for t in list(enc.parameters()) + [emb["seq.query"], emb["seq.sequence"]]:
                t.grad = None
            o = enc(emb)
            (o * torch.linspace(-1, 1, o.numel()).view_as(o)).sum().backward()
            outs[split] = [o.detach().clone()] + [t.grad.clone() for t in list(enc.parameters()) + [emb["seq.query"], emb["seq.sequence"]]]
        for a_, b_ in zip(outs[True], outs[False]):
            torch.testing.assert_close(a_, b_, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("case", ["plain", "narrow_query", "truncated", "max_seq_length", "wide_hidden", "wide_hidden_library", "own_narrow_query",
                                  "own_truncated"])
def test_din_jagged_positions_equal_the_padded_form(dev, case, monkeypatch):
    """DINEncoder.forward_jagged (csrc/din_attention.hip: the attention MLP on the rows of the unpooled lookup, softmax and
    weighted sum per sample's rows) against the reference's evaluation on the padded [B, L, D] tensor with the literal
    [q, k, q - k, q * k] input (/root/reference/tzrec/modules/sequence.py:101-128): output and EVERY gradient -- query, sequence
    rows, all attention-MLP parameters, the score layer -- elementwise to 1e-5.  Samples without any position, with one,
    with more than the padded length (truncated: rows behind it get zero gradient), `max_seq_length` below the padded
    length, a query narrower than the sequence rows, hidden widths that are no multiple of 16."""
    torch.manual_seed(3)
    rng = np.random.default_rng(7)
    D, qd, hidden, L, msl = 16, 16, [24, 8], 9, 0
    if case == "narrow_query":
        qd = 12
    if case == "max_seq_length":
        msl = 5
    from torcheasyrec_amd import dense
    from torcheasyrec_amd.sequence import _DinTowerFn

    # "wide_hidden" / "own_*": widths the library's own tall-input products take (csrc/gemm_rows.hip: the query's block of the first
    # layer once per sample) -- however few rows; "wide_hidden_library": the same widths through the GEMM library path
    own = case in ("wide_hidden", "own_narrow_query", "own_truncated")
    if case.startswith("wide_hidden") or own:
        D, qd, hidden = 48, 48, [256, 64]
    if case == "own_narrow_query":
        D, qd, hidden = 32, 20, [128, 64]
    monkeypatch.setattr(dense, "ROWS_GEMM_MIN_ROWS", 0)
    monkeypatch.setattr(dense, "OWN_ROWS_GEMM", case != "wide_hidden_library")
    calls0 = _DinTowerFn.own_calls
    B = 23
    lens = rng.integers(0, L + 1, size=B)
    lens[[0, 7]] = 0
    lens[3] = 1
    if case in ("truncated", "own_truncated"):
        lens[[2, 11]] = [L + 4, L + 1]  # longer than the padded length
    lens = lens.astype(np.int64)
    N = int(lens.sum())
    off = torch.zeros(B + 1, dtype=torch.int64)
    off[1:] = torch.from_numpy(np.cumsum(lens))
    enc = DINEncoder(D, qd, "seq", {"hidden_units": hidden}, max_seq_length=msl).to(dev)
    q0, v0 = torch.randn(B, qd), torch.randn(N, D)
    gw = torch.randn(B, D)
    res = {}
    for form in ("padded", "jagged"):
        for p_ in enc.parameters():
            p_.grad = None
        q, v = q0.clone().to(dev).requires_grad_(True), v0.clone().to(dev).requires_grad_(True)
        if form == "padded":
            enc.split_first_layer = False
            out = enc({"seq.query": q, "seq.sequence": jagged_to_padded_dense(v, off.to(dev), L),
                       "seq.sequence_length": torch.from_numpy(lens).to(dev)})
        else:
            assert enc.jagged_capable()
            enc.row_bucket = 1 if case == "plain" else 64  # (rows of the MLP input rounded up: zero rows behind the last position)
            out = enc({"seq.query": q, "seq.sequence_jagged": v, "seq.sequence_offsets": off.to(dev), "seq.sequence_max_len": L,
                       "seq.sequence_length": torch.from_numpy(lens).to(dev)})
        (out * gw.to(dev)).sum().backward()
        res[form] = [out.detach().cpu(), q.grad.cpu(), v.grad.cpu()] + [p_.grad.detach().cpu().clone() for p_ in enc.parameters()]
    names = ["out", "d query", "d sequence rows"] + [n for n, _ in enc.named_parameters()]
    for n, a_, b_ in zip(names, res["jagged"], res["padded"]):
        # (the score layer's bias gradient is sum_n ds_n = 0 exactly -- a softmax does not see a shift of its scores: both
        # forms return rounding noise of the order of 1e-6 there)
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-5 if n == "linear.bias" else 2e-6, msg=lambda m, n=n: f"{n}: {m}")
    assert (_DinTowerFn.own_calls > calls0) == own
    assert bool((res["jagged"][0][[0, 7]] == 0).all())  # no position: zero output (the reference: uniform weights over zero padding rows)
