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
