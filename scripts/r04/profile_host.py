"""Where the HOST time of the sharded step goes (it is the step's duration: profiles/r04o): cProfile over 300 steps of the
8192-per-rank step on a 1-rank RCCL group."""
import cProfile
import os
import pstats
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.distributed as dist


def main():
    from torcheasyrec_amd import _lib
    from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch
    from torcheasyrec_amd.dense import FusedDenseAdam
    from torcheasyrec_amd.embedding import SparseOptimizerConfig
    from torcheasyrec_amd.sharded_step import ShardedTrainStep
    from torcheasyrec_amd.sharding import ShardedDLRM

    _lib.use_native()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    work = torch.cuda.Stream(dev)
    torch.cuda.set_stream(work)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rows = [min(r, 2_000_000) for r in CRITEO_ROWS]
    B = 8192
    m = ShardedDLRM(criteo_tables(rows), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                    replicate_at_world1=True, exchange="capacity")
    ts = ShardedTrainStep(m, FusedDenseAdam(list(m.dense_parameters()), lr=1e-3), use_graph=True, step_graph=True)
    batches = [tuple(t.to(dev) for t in synthetic_batch(s, B, rows)) for s in range(8)]
    for i in range(16):
        ts.step(*batches[i % 8], next_kjt=batches[(i + 1) % 8][1])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(300):
        ts.step(*batches[i % 8], next_kjt=batches[(i + 1) % 8][1])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(22)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
