#!/usr/bin/env python3
"""Host-side profile of the 8 192-per-rank sharded step on a 1-rank RCCL group (bench.py's sharded proxy): cProfile over N steps
(the step is host-bound: where does the host's time go?).  `python scripts/r05/proxy_profile.py [steps] [--no-native]`"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, NUM_DENSE, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.dense import FusedDenseAdam  # noqa: E402
from torcheasyrec_amd.embedding import SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sharded_step import ShardedTrainStep  # noqa: E402
from torcheasyrec_amd.sharding import ShardedDLRM  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
native = "--no-native" not in sys.argv
_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
work = torch.cuda.Stream(dev)
torch.cuda.set_stream(work)
d = tempfile.mkdtemp()
dist.init_process_group("nccl", init_method=f"file://{d}/init", rank=0, world_size=1, device_id=dev)
B = 8192
m = ShardedDLRM(criteo_tables(CRITEO_ROWS), SPARSE_KEYS, NUM_DENSE, device=dev, sparse_optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                replicate_at_world1=True, exchange="capacity", dp_max_rows=65536)
ts = ShardedTrainStep(m, FusedDenseAdam(list(m.dense_parameters()), lr=1e-3), use_graph=True, step_graph=True, graph_input_dist=True,
                      native_driver=None if native else False)
batches = [tuple(t.to(dev) for t in synthetic_batch(s, B, CRITEO_ROWS)) for s in range(8)]


def run(n):
    for i in range(n):
        dn, k, lb = batches[i % 8]
        ts.step(dn, k, lb, next_kjt=batches[(i + 1) % 8][1])


run(24)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(steps)
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"native={native} steps={steps} ms_per_step={tot / steps * 1e3:.4f} host_queue_ms={host / steps * 1e3:.4f} native_steps={ts.native_steps}")
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print("\n".join(ln[:170] for ln in s.getvalue().splitlines()))
sys.stdout.flush()
os._exit(0)
