#!/usr/bin/env python3
"""Can RCCL calls on the library's OWN communicator (native_step.NativeComm) be captured into a hipGraph?  (With torch's
ProcessGroup they segfaulted / hung: NOTES.md round 2 / round 4.)  One case per process: `... probe.py <case>`;
cases: inline | forkjoin | child.  Prints PROBE_OK <case> on success."""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.native_step import NativeComm, StepProgram  # noqa: E402

case = sys.argv[1]
_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
work = torch.cuda.Stream(dev)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(work)
comm = NativeComm(None, dev)
print("comm created, RCCL", comm.version, flush=True)
a = torch.arange(1 << 16, dtype=torch.float32, device=dev)
b = torch.zeros_like(a)
c = torch.ones(1 << 14, dtype=torch.float32, device=dev)
# eager first (channel setup outside any capture)
comm.all_to_all(a, b)
comm.all_reduce(c)
torch.cuda.synchronize()
assert torch.equal(a, b) and float(c.sum()) == (1 << 14)
print("eager collectives ok", flush=True)
g = torch.cuda.CUDAGraph()
if case == "inline":
    with torch.cuda.graph(g, stream=work, capture_error_mode="thread_local"):
        a.add_(1.0)
        comm.all_to_all(a, b)
        c.mul_(2.0)
        comm.all_reduce(c)
        b.add_(0.5)
elif case == "forkjoin":
    e1, e2 = torch.cuda.Event(), torch.cuda.Event()
    with torch.cuda.graph(g, stream=work, capture_error_mode="thread_local"):
        a.add_(1.0)
        e1.record(work)
        side.wait_event(e1)
        comm.all_to_all(a, b, stream=side.cuda_stream)
        e2.record(side)
        c.mul_(2.0)            # independent work under the all-to-all
        work.wait_event(e2)
        comm.all_reduce(c)
        b.add_(0.5)
elif case == "child":
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=work, capture_error_mode="thread_local"):
        a.add_(1.0)
    with torch.cuda.graph(g2, stream=work, capture_error_mode="thread_local"):
        c.mul_(2.0)
        b.add_(0.5)
    P = StepProgram()
    P.add_graph(g1)
    x = P.add_all_to_all(comm, a, b)
    P.add_wait(x)
    P.add_graph(g2)
    y = P.add_all_reduce(comm, c)
    P.add_wait(y)
    with torch.cuda.graph(g, stream=work, capture_error_mode="thread_local"):
        P.run(work.cuda_stream)
else:
    raise SystemExit("case?")
print("captured", flush=True)
torch.cuda.synchronize()
a0, c0 = a.clone(), c.clone()
t0 = time.perf_counter()
n = 200
for _ in range(n):
    g.replay()
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"replays: host {host / n * 1e6:.1f} us, total {tot / n * 1e6:.1f} us per replay", flush=True)
assert torch.equal(a, a0 + n), (a[:4], a0[:4])
if case != "child":
    assert torch.allclose(b, a + 0.5)
print("PROBE_OK", case, flush=True)
os._exit(0)
