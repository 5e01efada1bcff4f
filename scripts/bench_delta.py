#!/usr/bin/env python3
"""Delta-embedding tracker at DLRM-Criteo scale: per-step cost of tzr_delta_mark next to the pooled forward
it shadows, and the dump-time cost of count + collect over all 204 M rows.

    python scripts/bench_delta.py            # on an MI355X (NOT run yet: written after round 1's GPU minutes were spent)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd import delta_embedding_dump as dd  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, SPARSE_KEYS, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, SparseOptimizerConfig  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    _lib.use_native()
    dev = torch.device("cuda", 0)
    B = 65536

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ebc = EmbeddingBagCollection(criteo_tables(CRITEO_ROWS), device=dev, optimizer=SparseOptimizerConfig(kind="adagrad", lr=0.01))

    m = M()
    batches = [synthetic_batch(s, B, CRITEO_ROWS)[1].to(dev) for s in range(4)]
    with torch.no_grad():
        base = timed(lambda: m.ebc(batches[0]))
    tr = dd.ModelDeltaTracker(m)
    with torch.no_grad():
        first = timed(lambda: m.ebc(batches[0]), n=1)  # bits not set yet: every id pays its atomic
        steady = timed(lambda: m.ebc(batches[0]))      # same batch again: plain reads only
        i = [0]

        def fresh():
            i[0] += 1
            m.ebc(batches[i[0] % 4])
        mixed = timed(fresh)
    N = len(SPARSE_KEYS) * B
    print(f"pooled forward alone {base * 1e3:7.1f} us | + tracker: cold bits {first * 1e3:7.1f} us, rotating 4 batches {mixed * 1e3:7.1f} us, "
          f"all bits set {steady * 1e3:7.1f} us  ({N} ids, {8 * N / 1e6:.1f} MB of ids)")
    torch.cuda.synchronize()
    import time

    t0 = time.perf_counter()
    ids = tr.get_unique_ids()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(v.numel() for v in ids.values())
    print(f"get_unique over {sum(CRITEO_ROWS) / 1e6:.0f} M rows ({sum(CRITEO_ROWS) / 8e6:.1f} MB of bitmaps): {dt * 1e3:.2f} ms, {n} touched rows")


if __name__ == "__main__":
    main()
