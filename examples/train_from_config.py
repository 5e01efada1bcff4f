#!/usr/bin/env python3
"""End to end on one MI355X from a tzrec text config: parse -> build the rank model (tables, groups,
fused sparse optimizer from the config) -> train on synthetic Criteo-shaped batches.

    python examples/train_from_config.py tests/golden/deepfm_mini.config

What a tzrec user keeps: the pipeline config, feature / group / model semantics, `pipeline.progress`.
What changes underneath: the embedding path runs on libtzrec_hip.so."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.config import load_pipeline_spec  # noqa: E402
from torcheasyrec_amd.dense import FusedDenseAdam  # noqa: E402
from torcheasyrec_amd.embedding_group import BASE_DATA_GROUP, Batch, TrainPipeline  # noqa: E402
from torcheasyrec_amd.lr_scheduler import create_scheduler  # noqa: E402
from torcheasyrec_amd.rank_model import build_rank_model  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor, KeyedTensor  # noqa: E402


def synthetic_batches(spec, n_rows, batch_size, seed=0):
    rng = np.random.default_rng(seed)
    sparse = [f for f in spec.features if f.is_sparse]
    dense = [f for f in spec.features if not f.is_sparse]
    for s in range(0, n_rows, batch_size):
        b = min(batch_size, n_rows - s)
        ids = np.concatenate([rng.integers(0, f.num_embeddings, size=b) for f in sparse]).astype(np.int64)
        kjt = KeyedJaggedTensor([f.name for f in sparse], torch.from_numpy(ids), torch.ones(len(sparse) * b, dtype=torch.int32),
                                uniform_length=1)
        kt = KeyedTensor([f.name for f in dense], [f.value_dim for f in dense],
                         torch.from_numpy(rng.random((b, sum(f.value_dim for f in dense)), dtype=np.float32)))
        yield Batch({BASE_DATA_GROUP: kt}, {BASE_DATA_GROUP: kjt}, {spec.label_fields[0]: torch.from_numpy((rng.random(b) < 0.25).astype(np.int64))})


def main(path):
    _lib.use_native()
    dev = torch.device("cuda", 0)
    spec = load_pipeline_spec(open(path).read())
    model = build_rank_model(spec, device=dev)
    opt = FusedDenseAdam(list(model.dense_parameters()), lr=spec.dense_lr)
    # the `learning_rate` oneof of the two optimizer blocks (tzrec/main.py:877-882); stepped per step
    # unless by_epoch (main.py:542-544)
    schedulers = [create_scheduler(model.fused_optimizer, spec.sparse_optimizer_block), create_scheduler(opt, spec.dense_optimizer_block)]
    # train_config.grad_clipping / gradient_accumulation_steps around the dense optimizer (tzrec/main.py:848-876)
    from torcheasyrec_amd.optimizer import build_train_optimizer

    pipe = TrainPipeline(model, build_train_optimizer(opt, spec.grad_clipping, spec.gradient_accumulation_steps), dev, model.loss)
    # train_config.delta_embedding_dump_config: the call sites of tzrec/main.py:805-811,900,547,611,928
    dumper = None
    if spec.delta_embedding_dump_config is not None:
        from torcheasyrec_amd.delta_embedding_dump import DeltaEmbeddingDumper

        dumper = DeltaEmbeddingDumper(model, spec.delta_embedding_dump_config, os.environ.get("MODEL_DIR", "experiments/model"), dev)
        dumper.start()
    it = iter(synthetic_batches(spec, 20 * (spec.batch_size or 1024), spec.batch_size or 1024))
    step = 0
    while True:
        try:
            losses, preds, _ = pipe.progress(it)
        except StopIteration:
            break
        step += 1
        for sch in schedulers:
            if not sch.by_epoch:
                sch.step()
        if dumper is not None:
            dumper.maybe_dump(step)
        if step % 5 == 0:
            print(f"step {step}: " + ", ".join(f"{k}={float(v.detach()):.4f}" for k, v in losses.items()))
    if dumper is not None:
        print("delta embedding dump:", dumper.final_dump(step))
        dumper.close()
    print("tables:", {n: tuple(w.shape) for n, w in model.embedding_group.ebc.table_weights().items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "deepfm_mini.config"))
