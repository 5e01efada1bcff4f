"""hipGraph captures next to a live RCCL process group, many times in ONE process.

Every iteration builds a ShardedTrainStep over a small sharded DLRM on a 1-rank RCCL group and runs it until both
pipeline slots have captured their graphs (6 graphs per slot + the input dist), each capture opened right behind eager
collectives -- the window in which the group's watchdog used to abort the process (hipEventQuery of a listed collective
while a capture is open).  `sharded_step._quiesce_process_group` closes it with a handshake through the flight recorder;
this script counts the captures, the seconds spent in that handshake, and ends with a regular destroy_process_group().

    python scripts/capture_stress.py [iterations=40]
"""
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch
import torch.distributed as dist


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    from torcheasyrec_amd import _lib, sharded_step
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
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)  # a SYNC collective on the stream that will capture: the case only the handshake covers
    import pickle

    from torch._C._distributed_c10d import _dump_nccl_trace
    ents = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries", [])
    print("flight recorder entries:", len(ents), "keys:", sorted(ents[0].keys()) if ents else None, flush=True)
    waits = []
    orig = sharded_step._quiesce_process_group

    def counted(device):
        w = orig(device)
        waits.append(w)
        return w

    sharded_step._quiesce_process_group = counted
    rows = [min(r, 20000) for r in CRITEO_ROWS]
    B, steps = 2048, 10
    batches = [tuple(t.to(dev) for t in synthetic_batch(s, B, rows)) for s in range(steps)]
    t0 = time.time()
    graphs = 0
    for it in range(iters):
        m = ShardedDLRM(criteo_tables(rows, init="seeded"), SPARSE_KEYS, NUM_DENSE, device=dev,
                        sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.05), dp_max_rows=4096, replicate_at_world1=True,
                        exchange="capacity")
        ts = ShardedTrainStep(m, FusedDenseAdam(list(m.dense_parameters()), lr=1e-2), use_graph=True, step_graph=True,
                              graph_input_dist=bool(it & 1))
        for s, (dense, kjt, label) in enumerate(batches):
            ts.step(dense, kjt, label, next_kjt=batches[s + 1][1] if s + 1 < steps else None)
        torch.cuda.synchronize()
        assert ts.graph_steps == steps and all(sl["graph"] is not None for sl in ts._slots.values())
        graphs += sum(len(sl["graph"]) + (2 if sl.get("in_graphs") else 0) for sl in ts._slots.values())
        del ts, m
        if it % 10 == 0:
            print(f"  iteration {it}: {graphs} graphs so far, recorder {sharded_step._FR_STATE['on']}, stuck {len(sharded_step._FR_STATE['stuck'])}, "
                  f"max wait {1e3 * max(waits):.1f} ms", flush=True)
        if it % 4 == 0:  # every few iterations a sync collective on the capturing stream right before the next captures
            dist.all_reduce(t)
    el = time.time() - t0
    rec = sharded_step._FR_STATE["on"]
    print(f"capture_stress: {iters} iterations, {graphs} graphs captured next to the process group in {el:.1f} s; "
          f"flight recorder {'on' if rec else 'OFF (timed fallback)'}; handshake waits: n = {len(waits)}, "
          f"mean {1e3 * sum(waits) / max(len(waits), 1):.1f} ms, max {1e3 * max(waits):.1f} ms", flush=True)
    dist.destroy_process_group()
    print("capture_stress: process group destroyed normally: OK", flush=True)


if __name__ == "__main__":
    main()
