#!/usr/bin/env python3
"""Embedding path only (pooled forward, backward plan, backward apply) on the DLRM-Criteo tables at
B=65536, for kernel traces: `rocprofv3 --kernel-trace -- python scripts/plan_trace.py`."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, SparseOptimizerConfig  # noqa: E402

_lib.use_library(_build.build())
if len(sys.argv) > 4:
    assert _lib.lib().tzr_tune(b"bwd_ch", int(sys.argv[4])) == 0
dev = torch.device("cuda", 0)
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind=(sys.argv[3] if len(sys.argv) > 3 else "adagrad"), lr=1e-3),
                             groups={"sparse": SPARSE_KEYS})
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
opt_kind = sys.argv[3] if len(sys.argv) > 3 else "adagrad"
batches = [synthetic_batch(s, B, CRITEO_ROWS, dist=dist)[1].to(dev) for s in range(4)]
g = torch.randn(B, 416, device=dev) * 1e-3
torch.cuda.synchronize()
for i in range(12):
    kjt = batches[i % 4]
    ebc._launch_forward(kjt, ("sparse",))
    ebc.plan_backward(kjt, ("sparse",))
    ebc._launch_backward(kjt, ("sparse",), [g])
torch.cuda.synchronize()
