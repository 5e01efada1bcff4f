#!/usr/bin/env python3
"""Pooled-forward experiments: where does the time go?  (tile size, table residency, id skew)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, SparseOptimizerConfig  # noqa: E402

_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
B = 65536


def timeit(ebc, kjts, n=20):
    for k in kjts:
        ebc._launch_forward(k, ("sparse",))
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        ebc._launch_forward(kjts[i % len(kjts)], ("sparse",))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for cap in (0, 40000):
    rows = [min(r, cap) for r in CRITEO_ROWS] if cap else list(CRITEO_ROWS)
    ebc = EmbeddingBagCollection(criteo_tables(rows), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3),
                                 groups={"sparse": SPARSE_KEYS})
    for dist in ("uniform", "zipf"):
        kjts = [synthetic_batch(s, B, rows, dist=dist)[1].to(dev) for s in range(4)]
        for G in (1, 8):
            _lib.check(_lib.lib().tzr_tune(b"fwd_slot_groups", G), "tune")
            res = []
            for tb in (0, 32, 64, 128, 256, 512):
                _lib.check(_lib.lib().tzr_tune(b"fwd_tile_b", tb), "tune")
                res.append(f"tile{tb}={timeit(ebc, kjts):.1f}")
            print(f"rows_cap={cap} ids={dist} groups={G}: " + " ".join(res), flush=True)
    del ebc
    torch.cuda.empty_cache()
