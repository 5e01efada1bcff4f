#!/usr/bin/env python3
"""INT8 row-wise export throughput: one 40 M x 16 fp32 table (2.56 GB) -> QUint8RowwiseF16 rows (0.8 GB).

    python scripts/bench_export.py            # on an MI355X
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib, export as ex  # noqa: E402


def main():
    _lib.use_native()
    dev = torch.device("cuda", 0)
    for rows, D in ((40_000_000, 16), (10_000_000, 64), (4_000_000, 128)):
        w = torch.randn(rows, D, device=dev)
        out = torch.empty(rows, D + 4, dtype=torch.uint8, device=dev)
        bad = torch.empty(3, dtype=torch.int64, device=dev)
        L = _lib.lib()

        def run():
            _lib.check(L.tzr_quantize_rows_q8f16(_lib.ptr(w), _lib.DT_F32, w.stride(0), rows, D, _lib.ptr(out), _lib.ptr(bad),
                                                 _lib.stream_ptr(dev)), "q")

        for _ in range(2):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            run()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        by = rows * (4 * D + D + 4)
        dq = ex.dequantize_quint8_rowwise_f16(out[:1000], D)
        err = float((dq - w[:1000]).abs().max())
        print(f"rows={rows:>10d} D={D:4d}  {ms:7.3f} ms  {by / ms / 1e9:6.2f} TB/s (read fp32 + write int8 rows)  max |dq - w| on 1000 rows {err:.4f}")
        del w, out


if __name__ == "__main__":
    main()
