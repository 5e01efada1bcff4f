#!/usr/bin/env python3
"""The same config on N GPUs of one node: tables sharded by the planner, dense parameters data-parallel.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \\
        examples/train_sharded.py tests/golden/din_mini.config [exact|capacity]

One process per GPU over RCCL (`backend="nccl"` is RCCL on ROCm).  Every rank reads its own slice of
the data (here: its own synthetic batches), as tzrec's per-rank `batch_size` means."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from examples.train_from_config import synthetic_batches  # noqa: E402
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.config import load_pipeline_spec  # noqa: E402
from torcheasyrec_amd.dense import FusedDenseAdam  # noqa: E402
from torcheasyrec_amd.embedding_group import TrainPipeline  # noqa: E402
from torcheasyrec_amd.planner import plan_to_json  # noqa: E402
from torcheasyrec_amd.rank_model import build_rank_model  # noqa: E402


def main(path, exchange="exact"):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    _lib.use_native()
    spec = load_pipeline_spec(open(path).read())
    bs = spec.batch_size or 1024
    # placement: the DP planner over the pooled tables of the config, under its `embedding_constraints` /
    # `global_embedding_constraints` if it has any (every rank computes the same plan; tzrec/main.py:783-799)
    # (`exchange="capacity"`: fixed-slice ids exchange, no split sizes through the host -- sharding.py)
    model = build_rank_model(spec, device=dev, process_group=dist.group.WORLD, use_planner=world > 1, exchange=exchange)
    ebc = model.embedding_group.ebc
    if rank == 0 and ebc is not None and world > 1:
        print(plan_to_json(ebc.sharding_plan() if hasattr(ebc, "sharding_plan") else ebc.plan()))
    opt = FusedDenseAdam(list(model.dense_parameters()), lr=spec.dense_lr)
    pipe = TrainPipeline(model, opt, dev, model.loss)
    it = iter(synthetic_batches(spec, 20 * bs, bs, seed=rank))
    step = 0
    while True:
        try:
            losses, _, _ = pipe.progress(it)
        except StopIteration:
            break
        step += 1
        if rank == 0 and step % 5 == 0:
            print(f"step {step}: " + ", ".join(f"{k}={float(v.detach()):.4f}" for k, v in losses.items()), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "din_mini.config"),
         sys.argv[2] if len(sys.argv) > 2 else "exact")
