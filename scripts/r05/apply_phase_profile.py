#!/usr/bin/env python3
"""Wall-clock phases of every workgroup of the planned apply (libtzrec_hip_prof.so, -DIT_PROF): when it started, when its unit
was staged in LDS, when each wave finished its tiles, when the wave ranges were stitched, when it ended.  100 MHz clock."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, SparseOptimizerConfig  # noqa: E402

_lib.use_library(os.path.join(ROOT, "torcheasyrec_amd", "libtzrec_hip_prof.so"))
L = _lib.lib()
dev = torch.device("cuda", 0)
B = 65536
ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                             groups={"sparse": SPARSE_KEYS})
ebc.plan_mode = "exact"  # (the four-launch plan's apply is what this script clocks; round 6's one-launch plan: scripts/r06/cells_phase_profile.py)
batches = [synthetic_batch(s, B, CRITEO_ROWS)[1].to(dev) for s in range(3)]
g = torch.randn(B, 416, device=dev) * 1e-3
for i in range(4):
    k = batches[i % 3]
    ebc._launch_forward(k, ("sparse",))
    ebc.plan_backward(k, ("sparse",))
    ebc._launch_backward(k, ("sparse",), [g])
torch.cuda.synchronize()
n = 1792
buf = (C.c_uint64 * (n * 8))()
fn = C.CDLL(os.path.join(ROOT, "torcheasyrec_amd", "libtzrec_hip_prof.so")).tzr_bwd_prof_dump
fn.restype = C.c_int
assert fn(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
start, staged, waves, stitched, end = us(a[:, 0]), us(a[:, 1]), us(a[:, 2:6]), us(a[:, 6]), us(a[:, 7])
q = lambda v: " ".join(f"{np.percentile(v, p):7.1f}" for p in (0, 10, 50, 90, 100))
print(f"{len(a)} workgroups; percentiles 0 / 10 / 50 / 90 / 100, us after the first workgroup's start")
print("start             ", q(start))
print("staged - start    ", q(staged - start))
print("tiles  - staged   ", q(waves.max(1) - staged), " (slowest wave)")
print("tiles  - staged   ", q(waves.min(1) - staged), " (fastest wave)")
print("stitch - tiles    ", q(stitched - waves.max(1)))
print("end    - stitch   ", q(end - stitched))
print("end               ", q(end))
