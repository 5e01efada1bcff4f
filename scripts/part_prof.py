#!/usr/bin/env python3
"""Phase timeline of the plan's partition pass (tzr_tune bwd_prof): per chunk workgroup the 100 MHz wall clock at
entry / geometry done / ids loaded / ranked / row published / slab written / table scan done.
    python scripts/part_prof.py [B] [fused 0|1]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, SparseOptimizerConfig  # noqa: E402

_lib.use_library(_build.build())
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                             groups={"sparse": SPARSE_KEYS})
batches = [synthetic_batch(s, B, CRITEO_ROWS)[1].to(dev) for s in range(3)]
g = torch.randn(B, 416, device=dev) * 1e-3
L.tzr_tune(b"bwd_prof", 1)
L.tzr_tune(b"fwd_plan_fuse", fused)
N = 26 * B
o8 = (ctypes.c_int64 * 8)()
assert L.tzr_pooled_bwd_plan_view(N, N, 26, 26, 16, o8) == 0
nch = int(o8[5])
for it in range(4):
    kjt = batches[it % 3]
    ebc._launch_forward(kjt, ("sparse",), with_plan=True)
    ws = ebc.plan_backward(kjt, ("sparse",))
    torch.cuda.synchronize()
    prof = ws[o8[7]:o8[7] + 64 * nch].view(torch.int64).view(nch, 8).cpu().numpy().astype(np.float64)
    cd = ws[o8[4]:o8[4] + 64 * nch].view(torch.int32).view(nch, 16).cpu().numpy()
    ebc._launch_backward(kjt, ("sparse",), [g])
    torch.cuda.synchronize()
    if it < 2:
        continue
    live = cd[:, 0] >= 0
    p = prof[live] * 0.01  # us
    t0 = p[:, 0].min()
    p = p - t0
    names = ["entry", "geometry", "ids loaded", "ranked", "row published", "slab written", "scan done"]
    print(f"B {B} fused {fused} iteration {it}: {live.sum()} chunks")
    for k, nm in enumerate(names[:6]):
        col = p[:, k]
        print(f"  {nm:14s} min {col.min():7.2f}  mean {col.mean():7.2f}  p90 {np.percentile(col, 90):7.2f}  max {col.max():7.2f} us after the first entry")
    d = np.diff(p[:, :6], axis=1)
    for k in range(5):
        print(f"  phase {names[k]:>14s} -> {names[k + 1]:14s} mean {d[:, k].mean():6.2f}  p90 {np.percentile(d[:, k], 90):6.2f}  max {d[:, k].max():6.2f} us")
    scan = p[:, 6] > 0
    sd = p[scan, 6] - p[scan, 5]
    print(f"  table scans: {scan.sum()}, duration mean {sd.mean():6.2f} max {sd.max():6.2f} us; last scan ends {p[scan, 6].max():7.2f} us after the first entry")
