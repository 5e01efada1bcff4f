#!/usr/bin/env python3
"""Which launches does ONE 1-rank RCCL collective cost?  (the sharded step's per-step `fillBufferAligned` /
`copyBuffer` launches: are they RCCL's or ours?)   rocprofv3 --kernel-trace --stats -- python scripts/probe_rccl_fills.py <what>"""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
what = sys.argv[1] if len(sys.argv) > 1 else "a2a"
n = 100
x = torch.arange(1 << 16, device=dev, dtype=torch.float32)
y = torch.empty_like(x)
torch.cuda.synchronize()
if what == "a2a":
    for _ in range(n):
        dist.all_to_all_single(y, x, [x.numel()], [x.numel()])
elif what == "a2a_even":
    for _ in range(n):
        dist.all_to_all_single(y, x)
elif what == "allreduce":
    for _ in range(n):
        dist.all_reduce(x, op=dist.ReduceOp.AVG)
elif what == "none":
    pass
torch.cuda.synchronize()
print(what, "done", flush=True)
os._exit(0) if not os.environ.get("ROCP_TOOL_LIBRARIES") else dist.destroy_process_group()
