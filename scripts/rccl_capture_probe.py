#!/usr/bin/env python3
"""Which RCCL collectives survive hipGraph capture on a 1-rank group here?  One subprocess per case (a failing
capture takes the process down).  python scripts/rccl_capture_probe.py"""
import os
import subprocess
import sys

CASES = ["a2a_sync", "a2a_async", "allreduce_sync", "allreduce_async", "allreduce_avg", "a2a_sync_relaxed", "kernels_only"]


def case(name):
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from torcheasyrec_amd.sharded_step import _quiesce_process_group

    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.arange(1 << 16, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)

    def body():
        if name.startswith("a2a_sync"):
            dist.all_to_all_single(y, x)
        elif name == "a2a_async":
            w = dist.all_to_all_single(y, x, async_op=True)
            y2 = x * 2
            w.wait()
            y.add_(y2)
        elif name == "allreduce_sync":
            dist.all_reduce(x)
        elif name == "allreduce_async":
            w = dist.all_reduce(x, async_op=True)
            w.wait()
        elif name == "allreduce_avg":
            dist.all_reduce(x, op=dist.ReduceOp.AVG)
        else:
            y.copy_(x * 3)

    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        for _ in range(3):
            body()
        torch.cuda.synchronize()
        # round 2's attempt died inside hipStreamEndCapture; round 4 found why captures next to a process group die at all
        # (the watchdog polling events of collectives whose stream is capturing): wait until it lists nothing
        waited = _quiesce_process_group(dev)
        g = torch.cuda.CUDAGraph()
        kw = {"capture_error_mode": "relaxed"} if name.endswith("relaxed") else {"capture_error_mode": "thread_local"}
        with torch.cuda.graph(g, stream=s, **kw):
            body()
        g.replay()
        g.replay()
        torch.cuda.synchronize()
    print(name, "OK", float(y.sum()), f"quiesce {1e3 * waited:.0f} ms", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=400)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            err = [ln for ln in r.stderr.splitlines() if "rror" in ln or "fault" in ln][:2]
            print(f"{c:18s} rc={r.returncode} {tail} {err}", flush=True)
