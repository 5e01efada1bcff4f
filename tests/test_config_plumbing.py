"""BASELINE config 1 (plumbing): a tzrec pipeline config in text format -> features / feature groups
-> tables (naming, WIDE `_wide` dim 4, shared `embedding_name`, `feature@table`) -> model -> train
steps through the pipeline; logits checked against the oracle's DeepFM restatement."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import emu_heavy  # noqa: E402
from oracle import tzrec_oracle as orc  # noqa: E402
from torcheasyrec_amd.config import load_pipeline_spec, parse_text_proto  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS  # noqa: E402
from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, EmbeddingGroup, TrainPipeline  # noqa: E402
from torcheasyrec_amd.rank_model import build_rank_model  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor  # noqa: E402

HERE = os.path.dirname(__file__)
REF = "/root/reference/examples"


def test_text_format_parser_basics():
    m = parse_text_proto('a { b: 1 c: "x\\ty" d: [1, 2, 3] e: FOO } a { b: 2.5 } # comment\nf: true')
    assert len(m.many("a")) == 2 and m.many("a")[0].one("b") == 1 and m.many("a")[0].one("c") == "x\ty"
    assert m.many("a")[0].many("d") == [1, 2, 3] and m.many("a")[0].one("e") == "FOO"
    assert m.many("a")[1].one("b") == 2.5 and m.one("f") is True


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_example_configs_parse():
    spec = load_pipeline_spec(open(os.path.join(REF, "dlrm_criteo.config")).read())
    assert spec.model_name == "dlrm" and spec.batch_size == 8192
    assert [f.num_embeddings for f in spec.features if f.is_sparse] == CRITEO_ROWS
    assert sum(1 for f in spec.features if not f.is_sparse) == 13
    assert spec.sparse_optimizer.kind == "adagrad" and abs(spec.sparse_optimizer.lr - 0.001) < 1e-9
    assert [g.group_name for g in spec.feature_groups] == ["dense", "sparse"]
    assert spec.model.one("arch_with_sparse") is True
    assert spec.model.one("final").many("hidden_units") == [64, 32]
    spec2 = load_pipeline_spec(open(os.path.join(REF, "deepfm_criteo.config")).read())
    assert spec2.model_name == "deepfm"
    assert [g.group_type for g in spec2.feature_groups] == ["WIDE", "DEEP", "DEEP"]
    assert spec2.model.one("deep").many("hidden_units") == [512, 256, 128]


def _batches(spec, n_rows, bs, seed=0):
    rng = np.random.default_rng(seed)
    sparse = [f for f in spec.features if f.is_sparse]
    dense = [f for f in spec.features if not f.is_sparse]
    ids = {f.name: rng.integers(0, f.num_embeddings, size=n_rows) for f in sparse}
    dv = {f.name: np.log(rng.integers(0, 1000, size=(n_rows, f.value_dim)) + 3.0).astype(np.float32) for f in dense}
    label = (rng.random(n_rows) < 0.25).astype(np.int64)
    for s in range(0, n_rows, bs):
        e = min(n_rows, s + bs)
        kjt = KeyedJaggedTensor([f.name for f in sparse], torch.from_numpy(np.concatenate([ids[f.name][s:e] for f in sparse])),
                                torch.ones(len(sparse) * (e - s), dtype=torch.int32))
        kt = KeyedTensor([f.name for f in dense], [f.value_dim for f in dense],
                         torch.from_numpy(np.concatenate([dv[f.name][s:e] for f in dense], axis=1)))
        yield Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt}, {"label": torch.from_numpy(label[s:e])})


def test_deepfm_config_to_training(dev):
    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "deepfm_mini.config")).read())
    torch.manual_seed(0)
    model = build_rank_model(spec, device=dev)
    eg = model.embedding_group
    # table construction rules (tzrec/modules/embedding.py:744-786, 576-600, 826-827)
    names = [c.name for c in eg.ebc.embedding_bag_configs()]
    assert names == ["cat_0_emb_wide", "cat_1_emb_wide", "cat_2_emb_wide", "cat_0_emb", "cat_1_emb", "cat_2_emb"]
    cfg = {c.name: c for c in eg.ebc.embedding_bag_configs()}
    assert cfg["cat_0_emb_wide"].embedding_dim == 4 and cfg["cat_0_emb"].embedding_dim == 16
    assert cfg["cat_2_emb"].feature_names == ["cat_2", "cat_3"]  # shared through embedding_name
    assert cfg["cat_2_emb_wide"].feature_names == ["cat_2", "cat_3"]
    assert eg.group_total_dim("wide") == 16 and eg.group_total_dim("fm") == 64 and eg.group_total_dim("deep") == 4 + 64
    assert eg.group_dims("deep") == [1, 1, 2, 16, 16, 16, 16]
    assert spec.sparse_optimizer.kind == "adagrad" and spec.batch_size == 250

    # first batch: logits vs the oracle restatement of DeepFM.predict
    first = next(_batches(spec, 1000, spec.batch_size))
    w = {n: t.detach().cpu().clone() for n, t in eg.ebc.table_weights().items()}
    with torch.no_grad():
        logits = model(first.to(dev))["logits"].cpu()
    kjt = first.sparse_features[BASE_DATA_GROUP]
    B = kjt.stride()
    deep_tabs = [w["cat_0_emb"], w["cat_1_emb"], w["cat_2_emb"], w["cat_2_emb"]]
    wide_tabs = [w["cat_0_emb_wide"], w["cat_1_emb_wide"], w["cat_2_emb_wide"], w["cat_2_emb_wide"]]
    bd = orc.pooled_lookup(deep_tabs, ["sum"] * 4, kjt.values(), kjt.lengths(), B)
    bw = orc.pooled_lookup(wide_tabs, ["sum"] * 4, kjt.values(), kjt.lengths(), B)
    lin = lambda seq: [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in seq if hasattr(m, "weight")]  # noqa: E731
    p = {"dim": 16, "deep_mlp": lin(model.deep_mlp.mlp), "final_mlp": lin(model.final_mlp.mlp),
         "output": (model.output_mlp.weight.detach().cpu(), model.output_mlp.bias.detach().cpu())}
    emb = torch.cat(bd, dim=1)
    dense = first.dense_features[BASE_DATA_GROUP].values()
    ref = orc.deepfm_forward(torch.cat(bw, dim=1), emb, torch.cat([dense, emb], dim=1), p)
    torch.testing.assert_close(logits, ref, rtol=1e-5, atol=1e-5)

    # 1k synthetic rows through pipeline.progress(): losses finite, tables move, StopIteration at end
    opt = torch.optim.Adam(list(model.dense_parameters()), lr=spec.dense_lr)
    pipe = TrainPipeline(model, opt, dev, model.loss)
    it = iter(_batches(spec, 1000, spec.batch_size))
    seen = []
    while True:
        try:
            losses, preds, batch = pipe.progress(it)
        except StopIteration:
            break
        seen.append(float(losses["binary_cross_entropy"]))
        assert preds["probs"].shape == (batch.labels["label"].shape[0],)
    assert len(seen) == 4 and all(np.isfinite(seen))
    assert not torch.equal(eg.ebc.table_weights()["cat_0_emb"].detach().cpu(), w["cat_0_emb"])
    # sparse LR schedulers mutate fused_optimizer.param_groups (tzrec/main.py:877-879)
    model.fused_optimizer.param_groups[0]["lr"] = 0.01
    assert abs(float(model.fused_optimizer.lr_device(dev).item()) - 0.01) < 1e-9


def test_zch_and_frozen_features_from_config(dev):
    """`zch {...}` and `trainable: false` in a feature config reach the kernels: raw 64-bit ids are
    remapped into zch_size rows before the lookup, the frozen table never moves."""
    text = open(os.path.join(HERE, "golden", "deepfm_mini.config")).read()
    text = text.replace('feature_configs { id_feature { feature_name: "cat_0" num_buckets: 1000 embedding_dim: 16 } }',
                        'feature_configs { id_feature { feature_name: "cat_0" embedding_dim: 16 '
                        'zch { zch_size: 64 eviction_interval: 2 distance_lfu { decay_exponent: 1.0 } '
                        'threshold_filtering_func: "lambda x: dynamic_threshold_filter(x, 0.0)" } } }')
    text = text.replace('feature_configs { id_feature { feature_name: "cat_1" num_buckets: 3 embedding_dim: 16 } }',
                        'feature_configs { id_feature { feature_name: "cat_1" num_buckets: 3 embedding_dim: 16 trainable: false } }')
    spec = load_pipeline_spec(text)
    f0 = next(f for f in spec.features if f.name == "cat_0")
    assert f0.num_embeddings == 64 and f0.zch is not None and not next(f for f in spec.features if f.name == "cat_1").trainable
    torch.manual_seed(0)
    model = build_rank_model(spec, device=dev)
    eg = model.embedding_group
    assert eg.mc is not None and set(eg.mc.modules_by_table) == {"cat_0_emb_wide", "cat_0_emb"}
    assert eg.mc.modules_by_table["cat_0_emb"].cfg.policy == "distance_lfu"
    frozen_before = {n: eg.ebc.table_weights()[n].detach().clone() for n in ("cat_1_emb", "cat_1_emb_wide")}
    opt = torch.optim.Adam(list(model.dense_parameters()), lr=spec.dense_lr)
    pipe = TrainPipeline(model, opt, dev, model.loss)

    def batches():
        for b in _batches(spec, 1000, spec.batch_size):  # cat_0 carries raw ids far outside any table
            kjt = b.sparse_features[BASE_DATA_GROUP]
            v = kjt.values().clone()
            n0 = int(kjt.lengths()[:kjt.stride()].sum())
            v[:n0] = (v[:n0] % 40) * 1_000_003 + (1 << 40)
            b.sparse_features[BASE_DATA_GROUP] = KeyedJaggedTensor(kjt.keys(), v, kjt.lengths())
            yield b

    it = iter(batches())
    n = 0
    while True:
        try:
            losses, _, _ = pipe.progress(it)
        except StopIteration:
            break
        assert np.isfinite(float(losses["binary_cross_entropy"]))
        n += 1
    assert n == 4
    m = eg.mc.modules_by_table["cat_0_emb"]
    held = m.row_ids[m.row_ids != (1 << 63) - 1]
    assert 0 < held.numel() <= 40 and bool((held >= (1 << 40)).all())  # 40 distinct raw ids own rows now
    for n_, w in frozen_before.items():
        assert torch.equal(eg.ebc.table_weights()[n_].detach(), w), n_


def _din_batches(spec, n_rows, batch_size, seed=0):
    rng = np.random.default_rng(seed)
    sparse = [f for f in spec.features if f.is_sparse]
    dense = [f for f in spec.features if not f.is_sparse]
    for s in range(0, n_rows, batch_size):
        b = min(batch_size, n_rows - s)
        vals, lens = [], []
        for f in sparse:
            ln = rng.integers(0, f.sequence_length + 3, size=b).astype(np.int32) if f.is_sequence else np.ones(b, np.int32)
            if f.is_sequence:
                ln[0] = 0  # an empty history
            lens.append(ln)
            vals.append(rng.integers(0, f.num_embeddings, size=int(ln.sum())))
        # the sequence sub-features of one sequence_feature share their lengths
        seq = [i for i, f in enumerate(sparse) if f.is_sequence]
        for i in seq[1:]:
            lens[i] = lens[seq[0]]
            vals[i] = rng.integers(0, sparse[i].num_embeddings, size=int(lens[i].sum()))
        kjt = KeyedJaggedTensor([f.name for f in sparse], torch.from_numpy(np.concatenate(vals).astype(np.int64)),
                                torch.from_numpy(np.concatenate(lens)))
        kt = KeyedTensor([f.name for f in dense], [f.value_dim for f in dense],
                         torch.from_numpy(rng.random((b, sum(f.value_dim for f in dense)), dtype=np.float32)))
        yield Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt}, {"clk": torch.from_numpy((rng.random(b) < 0.3).astype(np.int64))})


def test_multi_tower_din_config_to_training(dev):
    """BASELINE config 4 at the config level: `sequence_feature` blocks, a bucketized raw feature, a
    DEEP and a SEQUENCE group, `multi_tower_din`.  Logits against the oracle restatement (pooled
    lookup + per-id rows padded to the batch's longest history + DIN attention + MLPs), then training."""
    emu_heavy(dev)
    ref_cfg = os.path.join(REF, "multi_tower_din_taobao.config")
    if os.path.exists(ref_cfg):  # the reference's own example parses into the same structures
        big = load_pipeline_spec(open(ref_cfg).read())
        assert big.model_name == "multi_tower_din" and len(big.features) == 19
        seqf = [f for f in big.features if f.is_sequence]
        assert [f.name for f in seqf] == ["click_50_seq__adgroup_id", "click_50_seq__cate_id", "click_50_seq__brand"]
        assert all(f.sequence_length == 100 and f.embedding_dim == 16 for f in seqf)
        price = next(f for f in big.features if f.name == "price")
        assert price.is_sparse and price.num_embeddings == 99  # 98 boundaries + 1 (raw_feature.py:50-60)
        assert [g.group_type for g in big.feature_groups] == ["DEEP", "SEQUENCE"]
    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "din_mini.config")).read())
    torch.manual_seed(0)
    model = build_rank_model(spec, device=dev)
    eg = model.embedding_group
    assert [c.name for c in eg.ebc.embedding_bag_configs()] == ["user_id_emb", "adgroup_id_emb", "cate_id_emb", "price_emb"]
    ec = eg.ecs["16"]
    # the sequence group has its OWN tables, also for features that sit in the deep group too
    assert set(ec.table_weights()) == {"adgroup_id_emb", "cate_id_emb", "click_seq__adgroup_id_emb", "click_seq__cate_id_emb"}
    assert ec.table_weights()["adgroup_id_emb"].data_ptr() != eg.ebc.table_weights()["adgroup_id_emb"].data_ptr()
    assert eg.group_total_dim("deep") == 4 * 16 + 1 and eg.group_total_dim("seq.query") == 32 == eg.group_total_dim("seq.sequence")

    first = next(_din_batches(spec, 32, 32))
    with torch.no_grad():
        logits = model(first.to(dev))["logits"].cpu()
    kjt = first.sparse_features[BASE_DATA_GROUP]
    B = kjt.stride()
    off = orc.lengths_to_offsets(kjt.lengths().numpy())
    key = {k: i for i, k in enumerate(kjt.keys())}
    ids = lambda k: kjt.values()[off[key[k] * B]:off[(key[k] + 1) * B]]  # noqa: E731
    lens = lambda k: kjt.lengths()[key[k] * B:(key[k] + 1) * B].to(torch.int64)  # noqa: E731
    wb = {n: t.detach().cpu() for n, t in eg.ebc.table_weights().items()}
    wc = {n: t.detach().cpu() for n, t in ec.table_weights().items()}
    dense = first.dense_features[BASE_DATA_GROUP].values()
    deep = torch.cat([wb["user_id_emb"][ids("user_id")], wb["adgroup_id_emb"][ids("adgroup_id")], wb["cate_id_emb"][ids("cate_id")],
                      wb["price_emb"][ids("price")], dense], dim=1)
    query = torch.cat([wc["adgroup_id_emb"][ids("adgroup_id")], wc["cate_id_emb"][ids("cate_id")]], dim=1)
    sl = lens("click_seq__adgroup_id")
    lmax = min(int(sl.max()), 12)
    seq = torch.cat([orc.jagged_to_padded_dense(wc[f"{k}_emb"][ids(k)], lens(k), lmax) for k in ("click_seq__adgroup_id", "click_seq__cate_id")], dim=-1)
    lin = lambda seqm: [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in seqm if hasattr(m, "weight")]  # noqa: E731
    din = model.din_towers[0]
    y = torch.cat([orc.mlp(deep, lin(model.towers["deep"].mlp)),
                   orc.din_encoder(query, seq, sl, lin(din.mlp.mlp), (din.linear.weight.detach().cpu(), din.linear.bias.detach().cpu()))], dim=-1)
    y = orc.mlp(y, lin(model.final_mlp.mlp))
    ref = torch.nn.functional.linear(y, model.output_mlp.weight.detach().cpu(), model.output_mlp.bias.detach().cpu()).squeeze(1)
    torch.testing.assert_close(logits, ref, rtol=1e-5, atol=1e-5)

    opt = torch.optim.Adam(list(model.dense_parameters()), lr=spec.dense_lr)
    pipe = TrainPipeline(model, opt, dev, model.loss)
    it = iter(_din_batches(spec, 96, 32, seed=1))
    before = {n: t.detach().clone() for n, t in ec.table_weights().items()}
    n = 0
    while True:
        try:
            losses, preds, _ = pipe.progress(it)
        except StopIteration:
            break
        assert np.isfinite(float(losses["binary_cross_entropy"].detach()))
        n += 1
    assert n == 3
    for name, w in before.items():  # the unpooled tables were trained through the fused optimizer
        assert not torch.equal(ec.table_weights()[name].detach(), w), name


def test_multi_tower_din_jagged_step_against_the_oracle(dev):
    """The default config-4 model (DIN tower on the jagged rows: csrc/din_attention.hip, linear_bwd.hip) differentiated END TO END
    against the oracle's autograd on the padded restatement (oracle/tzrec_oracle.din_encoder = tzrec/modules/sequence.py:101-128,
    pinned to the reference module's outputs AND parameter gradients by tests/test_reference_module_vectors.py): loss, the gradient
    of EVERY dense parameter (deep tower, attention MLP, score layer, final MLP, output layer) elementwise 1e-5, and every
    table of both collections after the step's fused Adagrad update (pooled deep-group tables and the sequence group's own)."""
    emu_heavy(dev)
    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "din_mini.config")).read())
    torch.manual_seed(0)
    model = build_rank_model(spec, device=dev)
    eg = model.embedding_group
    assert eg.jagged_sequence_groups == {"seq"}
    ec = eg.ecs["16"]
    first = next(_din_batches(spec, 48, 48, seed=9))
    kjt = first.sparse_features[BASE_DATA_GROUP]
    B = kjt.stride()
    off = orc.lengths_to_offsets(kjt.lengths().numpy())
    key = {k: i for i, k in enumerate(kjt.keys())}
    ids = lambda k: kjt.values()[off[key[k] * B]:off[(key[k] + 1) * B]]  # noqa: E731
    lens = lambda k: kjt.lengths()[key[k] * B:(key[k] + 1) * B].to(torch.int64)  # noqa: E731
    assert int(lens("click_seq__adgroup_id").min()) == 0 and int(lens("click_seq__adgroup_id").max()) > 12  # empty and truncated histories
    # ---- the oracle, on leaf copies of every parameter and table
    wb = {n: t.detach().cpu().clone().requires_grad_(True) for n, t in eg.ebc.table_weights().items()}
    wc = {n: t.detach().cpu().clone().requires_grad_(True) for n, t in ec.table_weights().items()}
    dense_named = [(n, p_) for n, p_ in model.named_parameters() if any(p_ is q for q in model.dense_parameters())]
    leaf = {n: p_.detach().cpu().clone().requires_grad_(True) for n, p_ in dense_named}
    by_obj = {id(p_): leaf[n] for n, p_ in dense_named}
    lin = lambda seqm: [(by_obj[id(m.weight)], by_obj[id(m.bias)]) for m in seqm if hasattr(m, "weight")]  # noqa: E731
    dense = first.dense_features[BASE_DATA_GROUP].values()
    deep = torch.cat([wb["user_id_emb"][ids("user_id")], wb["adgroup_id_emb"][ids("adgroup_id")], wb["cate_id_emb"][ids("cate_id")],
                      wb["price_emb"][ids("price")], dense], dim=1)
    query = torch.cat([wc["adgroup_id_emb"][ids("adgroup_id")], wc["cate_id_emb"][ids("cate_id")]], dim=1)
    sl = lens("click_seq__adgroup_id")
    seq = torch.cat([orc.jagged_to_padded_dense(wc[f"{k}_emb"][ids(k)], lens(k), 12) for k in ("click_seq__adgroup_id", "click_seq__cate_id")], dim=-1)
    din = model.din_towers[0]
    y = torch.cat([orc.mlp(deep, lin(model.towers["deep"].mlp)),
                   orc.din_encoder(query, seq, sl, lin(din.mlp.mlp), (by_obj[id(din.linear.weight)], by_obj[id(din.linear.bias)]))], dim=-1)
    y = orc.mlp(y, lin(model.final_mlp.mlp))
    ref_logits = torch.nn.functional.linear(y, by_obj[id(model.output_mlp.weight)], by_obj[id(model.output_mlp.bias)]).squeeze(1)
    ref_loss = orc.bce_with_logits(ref_logits, first.labels["clk"])
    ref_loss.backward()
    # ---- the product: one training step's forward + backward (the tables update inside it)
    batch = first.to(dev)
    pred = model(batch)
    loss = sum(model.loss(pred, batch).values())
    loss.backward()
    torch.testing.assert_close(pred["logits"].detach().cpu(), ref_logits.detach(), rtol=1e-5, atol=1e-5)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-5 * abs(float(ref_loss.detach()))
    for n, p_ in dense_named:
        assert leaf[n].grad is not None and p_.grad is not None, n
        torch.testing.assert_close(p_.grad.detach().cpu(), leaf[n].grad, rtol=1e-5, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")
    lr, eps = spec.sparse_optimizer.lr, 1e-8
    assert spec.sparse_optimizer.kind == "adagrad" and lr == 0.05
    for col, ws in ((eg.ebc, wb), (ec, wc)):
        for n, w in ws.items():
            g = w.grad if w.grad is not None else torch.zeros_like(w)
            touched = (g != 0).any(dim=1, keepdim=True)
            # Adagrad from a zero state: state = g^2, w -= lr g / (|g| + eps) on the rows looked up (optim/optimizer_builder.py:53-59)
            exp = torch.where(touched, w.detach() - lr * g / (g.abs() + eps), w.detach())
            # (the first Adagrad step moves a weight by ~lr * sign(g): where a gradient element nearly cancels, its rounding shows)
            torch.testing.assert_close(col.table_weights()[n].detach().cpu(), exp, rtol=1e-5, atol=5e-6, msg=lambda m, n=n: f"table {n}: {m}")


def test_static_sequence_padding_changes_shapes_not_results(dev):
    """`EmbeddingGroup.static_sequence_padding`: the sequence group padded to its configured `sequence_length` (no read-back of
    the batch's longest sequence: the step becomes capturable) -- the same logits and the same gradients as padded to the
    batch maximum, because everything behind a sample's length is masked."""
    emu_heavy(dev)
    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "din_mini.config")).read())
    # a batch whose longest history is shorter than the configured 12 steps
    batch = next(b for sd in range(50) for b in _din_batches(spec, 3, 3, seed=sd)
                 if 0 < int(b.sparse_features[BASE_DATA_GROUP].lengths().max()) < 9).to(dev)
    outs = {}
    for static in (False, True):
        torch.manual_seed(0)
        model = build_rank_model(spec, device=dev)  # (a fresh model per mode: the backward below trains the tables)
        model.embedding_group.static_sequence_padding = static
        model.embedding_group.jagged_sequence_groups.clear()  # (this test is about the PADDED form of the group; the default is the next test)
        L = model.embedding_group(batch)["seq.sequence"].shape[1]
        pred = model(batch)
        losses = model.loss(pred, batch)
        sum(losses.values()).backward()
        outs[static] = (L, pred["logits"].detach().clone(), [p_.grad.detach().clone() for p_ in model.dense_parameters()])
    assert outs[True][0] == 12 and outs[False][0] < 9
    torch.testing.assert_close(outs[True][1], outs[False][1], rtol=1e-6, atol=1e-6)
    for a_, b_ in zip(outs[True][2], outs[False][2]):
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-6)


def test_multi_tower_din_on_jagged_positions_equals_the_padded_model(dev):
    """multi_tower_din built from its config: the DIN tower takes the sequence group as rows of the unpooled lookup
    (`EmbeddingGroup.jagged_sequence_groups`, the default: csrc/din_attention.hip) -- the same logits, dense gradients and
    table updates as the reference's padded evaluation of the same model."""
    emu_heavy(dev)
    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "din_mini.config")).read())
    batch = next(iter(_din_batches(spec, 6, 6, seed=4))).to(dev)  # (lengths 0 .. sequence_length + 2: empty and truncated histories)
    outs = {}
    for jagged in (True, False):
        torch.manual_seed(0)
        model = build_rank_model(spec, device=dev)
        assert model.embedding_group.jagged_sequence_groups == {"seq"}
        if not jagged:
            model.embedding_group.jagged_sequence_groups.clear()
        g = model.embedding_group(batch)
        assert ("seq.sequence_jagged" in g) == jagged and ("seq.sequence" in g) != jagged
        pred = model(batch)
        sum(model.loss(pred, batch).values()).backward()
        tables = {}
        for _, col in [("ebc", model.embedding_group.ebc)] + [(d, ec._store) for d, ec in model.embedding_group.ecs.items()]:
            tables.update({n: w.detach().cpu().clone() for n, w in col.table_weights().items()})
        outs[jagged] = (pred["logits"].detach().cpu(), [p_.grad.detach().cpu().clone() for p_ in model.dense_parameters()], tables)
    torch.testing.assert_close(outs[True][0], outs[False][0], rtol=1e-5, atol=1e-6)
    for a_, b_ in zip(outs[True][1], outs[False][1]):
        torch.testing.assert_close(a_, b_, rtol=1e-5, atol=2e-6)
    for n in outs[True][2]:
        # (the first Adagrad step moves a weight by ~lr * sign(g): where a gradient element nearly cancels, its rounding shows)
        torch.testing.assert_close(outs[True][2][n], outs[False][2][n], rtol=1e-5, atol=5e-6, msg=lambda m, n=n: f"table {n}: {m}")


def test_mmoe_with_zch_config_to_training(dev):
    """BASELINE config 5 at the config level: `mmoe {...}` over a group whose user id goes through a
    zero-collision hash (LFU eviction); two task towers, two labels, two losses."""
    ref_cfg = os.path.join(REF, "mmoe_taobao.config")
    if os.path.exists(ref_cfg):
        big = load_pipeline_spec(open(ref_cfg).read())
        assert big.model_name == "mmoe" and int(big.model.one("num_expert")) == 3
        assert [str(t.one("tower_name")) for t in big.model.many("task_towers")] == ["ctr", "cvr"]
        assert big.label_fields == ["clk", "buy"]
    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "mmoe_mini.config")).read())
    torch.manual_seed(0)
    model = build_rank_model(spec, device=dev)
    eg = model.embedding_group
    assert eg.mc is not None and eg.mc.modules_by_table["user_id_emb"].cfg.policy == "lfu"
    rng = np.random.default_rng(0)
    users = rng.integers(1 << 40, 1 << 50, size=90).astype(np.int64)  # raw 64-bit user ids

    def batches(n):
        for _ in range(n):
            b = 64
            ids = np.concatenate([users[np.minimum(rng.zipf(1.5, size=b), 89)], rng.integers(0, 300, size=b), rng.integers(0, 20, size=b)])
            kjt = KeyedJaggedTensor(["user_id", "adgroup_id", "pid"], torch.from_numpy(ids.astype(np.int64)), torch.ones(3 * b, dtype=torch.int32))
            kt = KeyedTensor(["price"], [1], torch.from_numpy(rng.random((b, 1), dtype=np.float32)))
            yield Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt},
                        {"clk": torch.from_numpy((rng.random(b) < 0.3).astype(np.int64)), "buy": torch.from_numpy((rng.random(b) < 0.1).astype(np.int64))})

    first = next(batches(1)).to(dev)
    model.eval()
    with torch.no_grad():
        p = model(first)
        x = eg(first)["all"]
    lin = lambda seqm: [(m.weight.detach().cpu(), m.bias.detach().cpu()) for m in seqm if hasattr(m, "weight")]  # noqa: E731
    xe = x.cpu()
    experts = torch.stack([orc.mlp(xe, lin(e.mlp)) for e in model.mmoe.expert_mlps], dim=1)
    for i, tower in enumerate(["ctr", "cvr"]):
        gate = torch.softmax(torch.nn.functional.linear(xe, model.mmoe.gate_finals[i].weight.detach().cpu(), model.mmoe.gate_finals[i].bias.detach().cpu()), dim=1)
        t_in = (gate.unsqueeze(2) * experts).sum(1)
        y = orc.mlp(t_in, lin(model.task_mlps[i].mlp))
        ref = torch.nn.functional.linear(y, model.task_outputs[i].weight.detach().cpu(), model.task_outputs[i].bias.detach().cpu()).squeeze(1)
        torch.testing.assert_close(p[f"logits_{tower}"].cpu(), ref, rtol=1e-5, atol=1e-5)
    model.train()
    opt = torch.optim.Adam(list(model.dense_parameters()), lr=spec.dense_lr)
    pipe = TrainPipeline(model, opt, dev, model.loss)
    it = iter(batches(6))
    n = 0
    while True:
        try:
            losses, preds, _ = pipe.progress(it)
        except StopIteration:
            break
        assert set(losses) == {"binary_cross_entropy_ctr", "binary_cross_entropy_cvr"}
        assert all(np.isfinite(float(v.detach())) for v in losses.values())
        n += 1
    assert n == 6
    m = eg.mc.modules_by_table["user_id_emb"]
    held = m.row_ids[m.row_ids != (1 << 63) - 1]
    assert held.numel() > 5 and bool((held >= (1 << 40)).all())  # raw user ids were admitted to rows


@pytest.mark.parametrize("tables_format", ["files", "dcp"])
def test_checkpoint_covers_pooled_and_sequence_tables(dev, tmp_path, tables_format):
    emu_heavy(dev)
    from torcheasyrec_amd.checkpoint import read_plan, restore_checkpoint, save_checkpoint

    spec = load_pipeline_spec(open(os.path.join(HERE, "golden", "din_mini.config")).read())
    torch.manual_seed(0)
    a = build_rank_model(spec, device=dev)
    opt = torch.optim.Adam(list(a.dense_parameters()), lr=spec.dense_lr)
    pipe = TrainPipeline(a, opt, dev, a.loss)
    it = iter(_din_batches(spec, 48, 24, seed=3))
    for _ in range(2):
        pipe.progress(it)
    save_checkpoint(str(tmp_path), a, opt, tables_format=tables_format)
    assert set(read_plan(str(tmp_path))) == {"embedding_group.ebc", "embedding_group.ecs.16"}
    if tables_format == "dcp":  # the containers name things by the reference's module paths (tzrec/modules/embedding.py:194-195,855,1193)
        import torch.distributed.checkpoint as dcp

        names = set(dcp.FileSystemReader(os.path.join(str(tmp_path), "model", "dcp")).read_metadata().state_dict_metadata)
        assert "model.embedding_group.emb_impls.__BASE__.ebc.embedding_bags.user_id_emb.weight" in names, sorted(names)[:6]
        assert "model.embedding_group.seq_emb_impls.__BASE__.ec_dict.16.embeddings.click_seq__adgroup_id_emb.weight" in names or any(
            k.startswith("model.embedding_group.seq_emb_impls.__BASE__.ec_dict.16.embeddings.") for k in names), sorted(names)[:12]
    torch.manual_seed(9)
    b = build_rank_model(spec, device=dev)
    restore_checkpoint(str(tmp_path), b)
    probe = next(_din_batches(spec, 24, 24, seed=8)).to(dev)
    a.eval(), b.eval()
    with torch.no_grad():
        assert torch.equal(a(probe)["logits"], b(probe)["logits"])
    for n, w in a.embedding_group.ecs["16"].table_weights().items():
        assert torch.equal(b.embedding_group.ecs["16"].table_weights()[n].detach(), w.detach()), n
        assert torch.equal(b.embedding_group.ecs["16"].table_states()[n].detach(), a.embedding_group.ecs["16"].table_states()[n].detach())


def test_multivalued_sequence_steps_are_pooled_per_step(dev):
    """A sequence sub-feature with value_dim != 1 holds several ids per STEP: the unpooled rows of a
    step are pooled (segment_reduce, tzrec/modules/embedding.py:1353-1366) before padding.  Columns ->
    parse_sequence_column -> DataParser.to_kjt / to_mulval_lengths -> EmbeddingGroup, against
    torch.segment_reduce on the tables; the fused update against autograd on dense tables."""
    from torcheasyrec_amd.data_parser import DataParser, parse_sequence_column, parse_sequence_dense_column, parse_sparse_column
    from torcheasyrec_amd.embedding import SparseOptimizerConfig

    spec = load_pipeline_spec("""
    feature_configs { id_feature { feature_name: "item" num_buckets: 11 embedding_dim: 8 } }
    feature_configs { sequence_feature { sequence_name: "hist" sequence_length: 4
        features { id_feature { feature_name: "tags" num_buckets: 13 embedding_dim: 8 value_dim: 0 pooling: "mean" } }
        features { id_feature { feature_name: "cat" num_buckets: 7 embedding_dim: 8 } }
        features { raw_feature { feature_name: "dwell" value_dim: 4 } } } }
    model_config { feature_groups { group_name: "seq" group_type: SEQUENCE
        feature_names: "item" feature_names: "hist__tags" feature_names: "hist__cat" feature_names: "hist__dwell" } }
    """)
    tags = next(f for f in spec.features if f.name == "hist__tags")
    assert (tags.value_dim, tags.pooling) == (0, "mean")
    assert next(f for f in spec.features if f.name == "hist__cat").value_dim == 1
    lr = 0.5
    eg = EmbeddingGroup(spec.features, spec.feature_groups, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="sgd", lr=lr))
    S = chr(3)
    cols = {
        "item": parse_sparse_column("item", [3, 5, 3, 9]),
        # steps per sample 3, 1, 5 (truncated to sequence_length 4), 2; ids per step 1-3
        "hist__tags": parse_sequence_column("hist__tags", [f"1{S}2;3;4{S}5{S}6", "7", f"1;1{S}1;2;3{S}12;5", f"0;9{S}10"]),
        "hist__cat": parse_sequence_column("hist__cat", ["1;2;3", "4", "1;1;2;3;5", "0;6"]),
    }
    parser = DataParser(["item", "hist__tags", "hist__cat"], sequence_keys=["hist__tags", "hist__cat"], sequence_mulval_keys=["hist__tags"])
    rng = np.random.default_rng(2)
    dwell_rows = [[[float(x) for x in rng.integers(-4, 5, size=4) / 4] for _ in range(n)] for n in (3, 1, 5, 2)]
    dwell = parse_sequence_dense_column("hist__dwell", dwell_rows, value_dim=4)
    batch = Batch({}, {BASE_DATA_GROUP: parser.to_kjt(cols)}, {}, {}, {BASE_DATA_GROUP: parser.to_mulval_lengths(cols)},
                  DataParser.to_sequence_dense({"hist__dwell": dwell})).to(dev)
    mv = batch.sequence_mulval_lengths[BASE_DATA_GROUP]
    assert mv.keys() == ["hist__tags"] and mv.values().tolist() == [2, 1, 3, 1, 1, 2, 1, 2, 1, 1, 2] and mv.lengths().tolist() == [3, 1, 5, 2]

    d = batch.to_dict()  # the reference's flat key names (datasets/utils.py:465-512)
    assert d["hist__tags.key_lengths"].tolist() == [2, 1, 3, 1, 1, 2, 1, 2, 1, 1, 2] and d["hist__tags.lengths"].tolist() == [3, 1, 5, 2]
    assert d["hist__cat.lengths"].tolist() == [3, 1, 5, 2] and d["item.values"].tolist() == [3, 5, 3, 9]
    assert tuple(d["hist__dwell.values"].shape) == (11, 4) and d["hist__dwell.lengths"].tolist() == [3, 1, 5, 2]

    ec = eg.ecs["8"]
    w0 = {n: t.detach().cpu().clone() for n, t in ec.table_weights().items()}
    out = eg(batch)
    assert out["seq.sequence_length"].tolist() == [3, 1, 5, 2]
    assert tuple(out["seq.sequence"].shape) == (4, 4, 20) and tuple(out["seq.query"].shape) == (4, 8)
    assert eg.group_total_dim("seq.sequence") == 20

    # reference composition on dense torch tables
    wt = {n: w.clone().requires_grad_(True) for n, w in w0.items()}
    tag_ids = torch.from_numpy(np.array(cols["hist__tags"].values))
    step = torch.nan_to_num(torch.segment_reduce(wt["hist__tags_emb"][tag_ids], "mean", lengths=torch.from_numpy(cols["hist__tags"].lengths.astype(np.int64))), nan=0.0)
    seq_len = torch.tensor([3, 1, 5, 2])
    ref_seq = torch.cat([orc.jagged_to_padded_dense(step, seq_len, 4),
                         orc.jagged_to_padded_dense(wt["hist__cat_emb"][torch.from_numpy(np.array(cols["hist__cat"].values))], seq_len, 4),
                         orc.jagged_to_padded_dense(torch.from_numpy(np.array(dwell.values)), seq_len, 4)], dim=-1)
    ref_q = wt["item_emb"][torch.from_numpy(np.array(cols["item"].values))]
    torch.testing.assert_close(out["seq.sequence"].detach().cpu(), ref_seq.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out["seq.query"].detach().cpu(), ref_q.detach(), rtol=0, atol=0)

    g = torch.Generator().manual_seed(4)
    gs, gq = torch.randn(ref_seq.shape, generator=g), torch.randn(ref_q.shape, generator=g)
    ((out["seq.sequence"] * gs.to(dev)).sum() + (out["seq.query"] * gq.to(dev)).sum()).backward()
    ((ref_seq * gs).sum() + (ref_q * gq).sum()).backward()
    for n, w in ec.table_weights().items():
        torch.testing.assert_close(w.detach().cpu(), w0[n] - lr * wt[n].grad, rtol=1e-5, atol=1e-6)

    # a batch without the per-step counts is refused, not silently treated as longer sequences
    with pytest.raises(KeyError, match="sequence_mulval_lengths"):
        eg(Batch({}, {BASE_DATA_GROUP: parser.to_kjt(cols)}, {}, {}, {}, DataParser.to_sequence_dense({"hist__dwell": dwell})).to(dev))
    with pytest.raises(KeyError, match="sequence_dense_features"):
        eg(Batch({}, {BASE_DATA_GROUP: parser.to_kjt(cols)}, {}, {}, {BASE_DATA_GROUP: parser.to_mulval_lengths(cols)}).to(dev))


def test_sparse_adam_from_config_and_checkpoint(dev, tmp_path):
    """`adam_optimizer` in `sparse_optimizer` (the one other kind the reference's configs use,
    protos/optimizer.proto:89-96): parsed, trains, and a checkpoint carries the step counter so the
    bias correction continues where it stopped."""
    emu_heavy(dev)
    from torcheasyrec_amd.checkpoint import restore_checkpoint, save_checkpoint

    text = open(os.path.join(HERE, "golden", "deepfm_mini.config")).read()
    assert "adagrad_optimizer" in text
    text = text.replace("adagrad_optimizer", "adam_optimizer", 1)
    spec = load_pipeline_spec(text)
    so = spec.sparse_optimizer
    assert so.kind == "adam" and (so.beta1, so.beta2) == (0.9, 0.999)
    spec.sparse_optimizer.beta1, spec.sparse_optimizer.beta2 = 0.7, 0.9

    def make():
        torch.manual_seed(0)
        return build_rank_model(spec, device=dev)

    def steps(model, n, seed):
        opt = torch.optim.Adam(list(model.dense_parameters()), lr=spec.dense_lr)
        pipe = TrainPipeline(model, opt, dev, model.loss)
        it = iter(_batches(spec, n * 16, 16, seed=seed))
        for _ in range(n):
            pipe.progress(it)

    a = make()
    steps(a, 3, seed=1)
    fo = a.embedding_group.ebc.fused_optimizer
    assert float(fo.adam_state(dev)[0]) == 3.0
    st = next(iter(a.embedding_group.ebc.table_states().values()))
    D = next(iter(a.embedding_group.ebc.table_weights().values())).shape[1]
    assert st.shape[1] == 2 * D and float(st[:, D:].abs().sum()) > 0  # exp_avg_sq moved
    save_checkpoint(str(tmp_path / "ck"), a)
    b = make()
    restore_checkpoint(str(tmp_path / "ck"), b)
    assert float(b.embedding_group.ebc.fused_optimizer.adam_state(dev)[0]) == 3.0
    steps(a, 2, seed=2)
    steps(b, 2, seed=2)
    for n, w in a.embedding_group.ebc.table_weights().items():
        assert torch.equal(w.detach(), b.embedding_group.ebc.table_weights()[n].detach()), n
