#!/usr/bin/env python3
"""Start / end wall clock of every workgroup of the backward plan's four launches (libtzrec_hip_prof.so, -DIT_PROF)."""
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
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                             groups={"sparse": SPARSE_KEYS})
batches = [synthetic_batch(s, B, CRITEO_ROWS, dist=dist)[1].to(dev) for s in range(3)]
g = torch.randn(B, 416, device=dev) * 1e-3
for i in range(5):
    k = batches[i % 3]
    ebc._launch_forward(k, ("sparse",))
    ebc.plan_backward(k, ("sparse",))
    ebc._launch_backward(k, ("sparse",), [g])
torch.cuda.synchronize()
W = 4096
buf = (C.c_uint64 * (4 * W * 2))()
fn = C.CDLL(so).tzr_plan_prof_dump
fn.restype = C.c_int
assert fn(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, W, 2).astype(np.float64)
t0 = min(a[k][:, 0][a[k][:, 0] > 0].min() for k in range(4) if (a[k][:, 0] > 0).any())
q = lambda v: " ".join(f"{np.percentile(v, p):7.1f}" for p in (0, 10, 50, 90, 100))
prev_end = None
for k, name in enumerate(("hist", "scan", "scatter", "sort")):
    x = a[k][a[k][:, 0] > 0]
    st, en = (x[:, 0] - t0) / 100, (x[:, 1] - t0) / 100
    en = np.where(en < st, st, en)
    gap = "" if prev_end is None else f"  (first start {st.min() - prev_end:5.1f} us after the previous launch's last end)"
    print(f"{name:8s} {len(x):5d} workgroups: starts {q(st)} | durations {q(en - st)} | ends {q(en)}{gap}")
    prev_end = en.max()
