#!/usr/bin/env python3
"""Wall-clock phases of every workgroup of the Criteo-shaped pooled forward (libtzrec_hip_prof.so, -DIT_PROF)."""
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

so = os.path.join(ROOT, "torcheasyrec_amd", "libtzrec_hip_prof.so")
_lib.use_library(so)
dev = torch.device("cuda", 0)
B = 65536
ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                             groups={"sparse": SPARSE_KEYS})
batches = [synthetic_batch(s, B, CRITEO_ROWS)[1].to(dev) for s in range(3)]
for i in range(5):
    ebc._launch_forward(batches[i % 3], ("sparse",))
torch.cuda.synchronize()
n = 2048
buf = (C.c_uint64 * (n * 4))()
fn = C.CDLL(so).tzr_fwd_prof_dump
fn.restype = C.c_int
assert fn(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
q = lambda v: " ".join(f"{np.percentile(v, p):7.1f}" for p in (0, 10, 50, 90, 100))
print(f"{len(a)} workgroups; percentiles 0 / 10 / 50 / 90 / 100 (us)")
print("start after the first    ", q(us(a[:, 0])))
print("slots resolved - start   ", q((a[:, 1] - a[:, 0]) / 100))
print("ids in LDS - resolved    ", q((a[:, 2] - a[:, 1]) / 100))
print("gathers + stores         ", q((a[:, 3] - a[:, 2]) / 100))
print("end after the first start", q(us(a[:, 3])))
