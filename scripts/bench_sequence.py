#!/usr/bin/env python3
"""Sequence path at the scale of examples/multi_tower_din_taobao.config: batch
8192, click sequences of up to 100 item ids (mean ~50), one shared 4.2 M-row item table of dim 32.
Times the unpooled lookup (tzr_rows_gather), jagged -> padded dense (K12) and the fused sparse update
through it (K6 + K7 with per-id gradient rows), with the bytes each must move."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.embedding import SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sequence import EmbeddingCollection, EmbeddingConfig, jagged_to_padded_dense  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402

_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
B, L, D, ROWS = 8192, 100, 32, 4_200_000
rng = np.random.default_rng(0)
lens_seq = rng.integers(0, L + 1, size=B).astype(np.int32)
lens = np.concatenate([np.ones(B, np.int32), lens_seq])
vals = rng.integers(0, ROWS, size=int(lens.sum())).astype(np.int64)
kjt = KeyedJaggedTensor(["item_id", "click_seq__item_id"], torch.from_numpy(vals), torch.from_numpy(lens)).to(dev)
ec = EmbeddingCollection([EmbeddingConfig("item_emb", D, ROWS, ["item_id", "click_seq__item_id"])], device=dev,
                         optimizer=SparseOptimizerConfig(kind="adagrad", lr=1e-3))
N = int(lens.sum())
Nseq = int(lens_seq.sum())
U = int(np.unique(vals).size)
g = torch.randn(N, D, device=dev) * 1e-3
off_seq = torch.zeros(B + 1, dtype=torch.int64, device=dev)
off_seq[1:] = torch.cumsum(torch.from_numpy(lens_seq.astype(np.int64)).to(dev), 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1e7))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = ec._launch_forward(kjt)
seq_rows = rows[B:]
t_lookup = timeit(lambda: ec._launch_forward(kjt))
t_pad = timeit(lambda: jagged_to_padded_dense(seq_rows, off_seq, L))
t_bwd = timeit(lambda: ec._launch_backward(kjt, g))
rb = D * 4
out = {
    "workload": f"B={B}, seq<= {L} (mean {Nseq / B:.1f}), table {ROWS} x {D} fp32, {N} ids, {U} distinct rows",
    "unpooled_lookup": {"us": t_lookup, "bytes": N * (8 + 2 * rb), "GBps": N * (8 + 2 * rb) / t_lookup / 1e3},
    "jagged_to_padded": {"us": t_pad, "bytes": Nseq * rb + B * L * rb, "GBps": (Nseq * rb + B * L * rb) / t_pad / 1e3},
    "sparse_backward": {"us": t_bwd, "bytes": N * (8 + rb) + U * 4 * rb, "GBps": (N * (8 + rb) + U * 4 * rb) / t_bwd / 1e3},
}
print(json.dumps(out))
