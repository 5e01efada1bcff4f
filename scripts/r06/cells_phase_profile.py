#!/usr/bin/env python3
"""Wall-clock phases of every workgroup of the cells apply (libtzrec_hip_prof.so, -DIT_PROF; 100 MHz clock): start, bounds known,
lookups in registers, sorted in LDS, each wave's tiles done, end -- by kind of unit."""
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

lib = os.path.join(ROOT, "torcheasyrec_amd", "libtzrec_hip_prof.so")
_lib.use_library(lib)
dev = torch.device("cuda", 0)
B = 65536
ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                             groups={"sparse": SPARSE_KEYS})
ebc.plan_mode = "cells"
batches = [synthetic_batch(s, B, CRITEO_ROWS)[1].to(dev) for s in range(3)]
g = torch.randn(B, 416, device=dev) * 1e-3
for i in range(4):
    k = batches[i % 3]
    ebc._launch_forward(k, ("sparse",))
    ebc.plan_backward(k, ("sparse",))
    ebc._launch_backward(k, ("sparse",), [g])
torch.cuda.synchronize()
n = 2048
buf = (C.c_uint64 * (n * 16))()
fn = C.CDLL(lib).tzr_cells_prof_dump
fn.restype = C.c_int
assert fn(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 16).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
q = lambda v: " ".join(f"{np.percentile(v, p):7.1f}" for p in (0, 10, 50, 90, 100))
print(f"{len(a)} unit workgroups; percentiles 0 / 10 / 50 / 90 / 100, us after the first workgroup's start")
for name, sel in (("all", np.ones(len(a), bool)), ("split rows", a[:, 10] > 0), ("others, n >= 900", (a[:, 10] == 0) & (a[:, 9] >= 900)),
                  ("others, n < 900", (a[:, 10] == 0) & (a[:, 9] < 900))):
    x = a[sel]
    if not len(x):
        continue
    start, bounds, regs, srt, waves, end = us(x[:, 0]), us(x[:, 1]), us(x[:, 2]), us(x[:, 3]), us(x[:, 4:8]), us(x[:, 8])
    print(f"-- {name}: {len(x)} units, lookups {q(x[:, 9])}")
    print("start              ", q(start))
    print("bounds - start     ", q(bounds - start))
    print("registers - bounds ", q(regs - bounds))
    print("sorted - registers ", q(srt - regs))
    print("tiles - sorted     ", q(waves.max(1) - srt), " (slowest wave)")
    print("end - tiles        ", q(end - waves.max(1)))
    print("end                ", q(end))
