"""GraphTrainPipeline (hipGraph replay per device slot + H2D of the next batch on a copy stream) walks the same
trajectory as the eager TrainPipeline on the same pinned host batches (tzrec/utils/dist_util.py:221-303)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
def test_graph_pipeline_matches_eager_pipeline():
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch
    from torcheasyrec_amd.dense import FusedDenseAdam
    from torcheasyrec_amd.dlrm import DLRM, bce_with_logits
    from torcheasyrec_amd.embedding import SparseOptimizerConfig
    from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, GraphTrainPipeline, TrainPipeline
    from torcheasyrec_amd.sparse import KeyedTensor

    _lib.use_native()
    dev = torch.device("cuda", 0)
    rows = [min(r, 30000) for r in CRITEO_ROWS]
    B, n_steps = 1024, 9
    host = []
    for s in range(n_steps):
        d, k, l = synthetic_batch(s, B, rows, dist="zipf" if s % 2 else "uniform")
        host.append(Batch({BASE_DATA_GROUP: KeyedTensor([f"int_{i}" for i in range(NUM_DENSE)], [1] * NUM_DENSE, d)},
                          {BASE_DATA_GROUP: k}, {"label": l}).pin_memory())

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.m = DLRM(criteo_tables(rows, init="seeded"), SPARSE_KEYS, NUM_DENSE, device=dev,
                          sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.05))

        def forward(self, b):
            return self.m(b.dense_features[BASE_DATA_GROUP].values(), b.sparse_features[BASE_DATA_GROUP])

    loss_of = lambda pred, b: {"bce": bce_with_logits(pred, b.labels["label"])}  # noqa: E731
    res = []
    work = torch.cuda.Stream(dev)
    with torch.cuda.stream(work):
        # "wire32": the same host batches with their ids narrowed to int32 for the trip across PCIe (Batch.narrow_ids),
        # widened on the device behind the copy: the trajectory must not notice
        narrow = [b_.narrow_ids().pin_memory() for b_ in (Batch({BASE_DATA_GROUP: hb.dense_features[BASE_DATA_GROUP]}, {BASE_DATA_GROUP: hb.sparse_features[BASE_DATA_GROUP]},
                                                                  dict(hb.labels)) for hb in host)]
        assert narrow[0].sparse_features[BASE_DATA_GROUP].wire_values().dtype == torch.int32
        for cls, kw in ((TrainPipeline, {}), (GraphTrainPipeline, {}), (GraphTrainPipeline, {"stage_first": True}), (TrainPipeline, {"fetch_first": False}),
                        (GraphTrainPipeline, {"wire32": True}), (TrainPipeline, {"wire32": True})):
            kw = dict(kw)
            batches_in = narrow if kw.pop("wire32", False) else host
            model = M()
            opt = FusedDenseAdam(list(model.m.dense_parameters()), lr=1e-2)
            pipe = cls(model, opt, dev, loss_of, **kw)
            it = iter(batches_in)
            losses = []
            while True:
                try:
                    l, _, _ = pipe.progress(it)
                except StopIteration:
                    break
                losses.append(float(l["bce"]))
            torch.cuda.synchronize()
            assert len(losses) == n_steps
            res.append((losses, [p.detach().clone() for p in model.m.dense_parameters()],
                        {n: w.detach().clone() for n, w in model.m.ebc.table_weights().items()}))
            if cls is GraphTrainPipeline:
                assert pipe._graphs[0] is not None and pipe._graphs[1] is not None  # the late steps were replays
    la, pa, wa = res[0]
    for lb, pb, wb in res[1:]:  # both orders of queueing the next batch's H2D
        torch.testing.assert_close(torch.tensor(lb), torch.tensor(la), rtol=1e-6, atol=1e-7)
        for a, b in zip(pa, pb):
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)
        for n in wa:
            torch.testing.assert_close(wb[n], wa[n], rtol=1e-5, atol=1e-6, msg=n)


@pytest.mark.gpu
def test_graph_pipeline_honours_learning_rate_changes_after_capture():
    """A scheduler changes `param_groups[..]["lr"]` of the fused sparse optimizer and of the dense optimizer AFTER the
    step graphs were captured (and once exactly on a capture step): the replayed graphs must read the new rates from
    the device scalars (`sync_learning_rates` before every replay; no `fill_` inside a capture) -- same trajectory
    as the eager pipeline under the same schedule (/root/reference/tzrec/main.py:877-879 mutates the rates per step)."""
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch
    from torcheasyrec_amd.dense import FusedDenseAdam
    from torcheasyrec_amd.dlrm import DLRM, bce_with_logits
    from torcheasyrec_amd.embedding import SparseOptimizerConfig
    from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, GraphTrainPipeline, TrainPipeline
    from torcheasyrec_amd.sparse import KeyedTensor

    _lib.use_native()
    dev = torch.device("cuda", 0)
    rows = [min(r, 20000) for r in CRITEO_ROWS]
    B, n_steps = 512, 12
    host = []
    for s in range(n_steps):
        d, k, l = synthetic_batch(s, B, rows)
        host.append(Batch({BASE_DATA_GROUP: KeyedTensor([f"int_{i}" for i in range(NUM_DENSE)], [1] * NUM_DENSE, d)},
                          {BASE_DATA_GROUP: k}, {"label": l}).pin_memory())

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(5)
            self.m = DLRM(criteo_tables(rows, init="seeded"), SPARSE_KEYS, NUM_DENSE, device=dev,
                          sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.05, initial_accumulator_value=0.1))

        def forward(self, b):
            return self.m(b.dense_features[BASE_DATA_GROUP].values(), b.sparse_features[BASE_DATA_GROUP])

    loss_of = lambda pred, b: {"bce": bce_with_logits(pred, b.labels["label"])}  # noqa: E731
    sparse_lr = lambda i: 0.05 * (0.5 ** (i // 2))  # noqa: E731  changes every second step, also on the capture steps
    dense_lr = lambda i: 1e-2 / (1 + i)  # noqa: E731
    res = []
    work = torch.cuda.Stream(dev)
    with torch.cuda.stream(work):
        for cls in (TrainPipeline, GraphTrainPipeline):
            model = M()
            opt = FusedDenseAdam(list(model.m.dense_parameters()), lr=dense_lr(0))
            pipe = cls(model, opt, dev, loss_of)
            it = iter(host)
            losses = []
            for i in range(n_steps):
                model.m.ebc.fused_optimizer.param_groups[0]["lr"] = sparse_lr(i)
                opt.param_groups[0]["lr"] = dense_lr(i)
                l, _, _ = pipe.progress(it)
                losses.append(float(l["bce"]))
            torch.cuda.synchronize()
            res.append((losses, [p.detach().clone() for p in model.m.dense_parameters()],
                        {n: w.detach().clone() for n, w in model.m.ebc.table_weights().items()}))
            if cls is GraphTrainPipeline:
                assert pipe._graphs[0] is not None and pipe._graphs[1] is not None
    (la, pa, wa), (lb, pb, wb) = res
    torch.testing.assert_close(torch.tensor(lb), torch.tensor(la), rtol=1e-6, atol=1e-7)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)
    for n in wa:
        torch.testing.assert_close(wb[n], wa[n], rtol=1e-5, atol=1e-6, msg=n)
    # and the schedule did matter: a frozen rate ends somewhere else
    assert sparse_lr(n_steps - 1) != sparse_lr(0)


@pytest.mark.gpu
def test_sequence_model_step_is_capturable_with_static_padding():
    """multi_tower_din (a SEQUENCE group with variable-length histories): with `static_sequence_padding` and the ids-per-key
    counted while the batch is still in host memory, a whole train step captures into a hipGraph; one replay leaves the same
    tables and dense weights as one eager step of an identically seeded model."""
    import numpy as np

    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.config import load_pipeline_spec
    from torcheasyrec_amd.dense import FusedDenseAdam
    from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, _backward_of_losses, _losses_and_predictions
    from torcheasyrec_amd.rank_model import build_rank_model
    from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor

    _lib.use_native()
    dev = torch.device("cuda", 0)
    spec = load_pipeline_spec(open(os.path.join(os.path.dirname(__file__), "golden", "din_mini.config")).read())
    B = 64
    rng = np.random.default_rng(4)
    sparse = [f for f in spec.features if f.is_sparse]
    dense = [f for f in spec.features if not f.is_sparse]

    def batch():
        lens = [rng.integers(0, f.sequence_length + 1, size=B).astype(np.int32) if f.is_sequence else np.ones(B, np.int32) for f in sparse]
        seq = [i for i, f in enumerate(sparse) if f.is_sequence]
        for i in seq[1:]:
            lens[i] = lens[seq[0]]
        vals = [rng.integers(0, f.num_embeddings, size=int(ln.sum())) for f, ln in zip(sparse, lens)]
        kjt = KeyedJaggedTensor([f.name for f in sparse], torch.from_numpy(np.concatenate(vals).astype(np.int64)),
                                torch.from_numpy(np.concatenate(lens)))
        kt = KeyedTensor([f.name for f in dense], [f.value_dim for f in dense],
                         torch.from_numpy(rng.random((B, sum(f.value_dim for f in dense)), dtype=np.float32)))
        return Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt}, {"clk": torch.from_numpy((rng.random(B) < 0.3).astype(np.int64))}).to(dev)

    warm, b = batch(), batch()
    assert b.sparse_features[BASE_DATA_GROUP]._length_per_key is not None  # counted on the host side of .to()
    stream = torch.cuda.Stream(device=dev)
    states = []
    with torch.cuda.stream(stream):
        for captured in (False, True):
            torch.manual_seed(3)
            model = build_rank_model(spec, device=dev)
            model.embedding_group.static_sequence_padding = True
            opt = FusedDenseAdam(list(model.dense_parameters()), lr=spec.dense_lr)

            def step(bb):
                opt.zero_grad(set_to_none=True)
                losses, _ = _losses_and_predictions(model, model.loss, bb)
                _backward_of_losses(losses)
                opt.step()

            step(warm)  # (allocations, permutation tensors, lazily built state: outside the capture)
            if captured:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    step(b)
                g.replay()
            else:
                step(b)
            torch.cuda.synchronize()
            eg = model.embedding_group
            states.append([t.detach().clone() for t in list(eg.ebc.table_weights().values()) + list(eg.ecs["16"].table_weights().values())]
                          + [p.detach().clone() for p in model.dense_parameters()])
    for a_, b_ in zip(*states):
        torch.testing.assert_close(a_, b_, rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_zero_collision_hash_step_replays_from_a_graph():
    """MMoE with the user id behind a zero-collision hash (BASELINE configs[4] at test size, eviction every 2 steps): through
    GraphTrainPipeline -- the ZCH wrapper in ring mode: device iteration counter, candidates in a device ring, admission /
    eviction rounds run between replays -- the same losses, dense weights, tables and id -> row maps as the eager pipeline
    with its per-step candidate lists."""
    import numpy as np

    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.config import load_pipeline_spec
    from torcheasyrec_amd.dense import FusedDenseAdam
    from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, GraphTrainPipeline, TrainPipeline
    from torcheasyrec_amd.rank_model import build_rank_model
    from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor

    _lib.use_native()
    dev = torch.device("cuda", 0)
    spec = load_pipeline_spec(open(os.path.join(os.path.dirname(__file__), "golden", "mmoe_mini.config")).read())
    rng = np.random.default_rng(0)
    users = rng.integers(1 << 40, 1 << 50, size=400).astype(np.int64)
    b, n_steps = 64, 11
    host = []
    for _ in range(n_steps):
        ids = np.concatenate([users[np.minimum(rng.zipf(1.3, size=b), 399)], rng.integers(0, 300, size=b), rng.integers(0, 20, size=b)])
        kjt = KeyedJaggedTensor(["user_id", "adgroup_id", "pid"], torch.from_numpy(ids.astype(np.int64)), torch.ones(3 * b, dtype=torch.int32),
                                uniform_length=1)
        kt = KeyedTensor(["price"], [1], torch.from_numpy(rng.random((b, 1), dtype=np.float32)))
        host.append(Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt},
                          {"clk": torch.from_numpy((rng.random(b) < 0.3).astype(np.int64)), "buy": torch.from_numpy((rng.random(b) < 0.1).astype(np.int64))}).pin_memory())
    res = []
    work = torch.cuda.Stream(dev)
    with torch.cuda.stream(work):
        for cls in (TrainPipeline, GraphTrainPipeline):
            torch.manual_seed(0)
            model = build_rank_model(spec, device=dev)
            model.train()
            opt = FusedDenseAdam(list(model.dense_parameters()), lr=spec.dense_lr)
            pipe = cls(model, opt, dev, model.loss)
            it = iter(host)
            losses = []
            while True:
                try:
                    l, _, _ = pipe.progress(it)
                except StopIteration:
                    break
                losses.append([float(v) for _, v in sorted(l.items())])
            torch.cuda.synchronize()
            mc = model.embedding_group.mc
            m = mc.modules_by_table["user_id_emb"]
            assert mc._iter == n_steps and (cls is TrainPipeline or (mc.device_profile and int(mc._d_iter.item()) == n_steps))
            res.append((losses, [p.detach().clone() for p in model.dense_parameters()],
                        {n: w.detach().clone() for n, w in model.embedding_group.ebc.table_weights().items()},
                        (m.row_ids.clone(), m.counts.clone(), m.last_iter.clone())))
            if cls is GraphTrainPipeline:
                assert pipe._graphs[0] is not None and pipe._graphs[1] is not None
    (la, pa, wa, za), (lb, pb, wb, zb) = res
    torch.testing.assert_close(torch.tensor(lb), torch.tensor(la), rtol=1e-6, atol=1e-7)
    for a, b_ in zip(pa, pb):
        torch.testing.assert_close(b_, a, rtol=1e-5, atol=1e-6)
    for n in wa:
        torch.testing.assert_close(wb[n], wa[n], rtol=1e-5, atol=1e-6, msg=n)
    for a, b_ in zip(za, zb):
        assert torch.equal(a, b_)  # the maps: integer work
    assert int((za[0] != (1 << 63) - 1).sum()) > 5  # users were admitted
