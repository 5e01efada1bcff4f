"""Generate tests/golden/reference_quant_vectors.npz by RUNNING the reference's INT8 row encoder
(/root/reference/tzrec/utils/quant_util.py:25-196; plain numpy).  Authoring container only:

    python tests/golden/make_reference_quant_vectors.py
"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_reference_module_vectors as mk  # noqa: E402


def main():
    mk.install_reference_imports()
    Q = importlib.import_module("tzrec.utils.quant_util")
    rng = np.random.default_rng(20260925)
    out = {}
    for D in (4, 16, 12, 64, 128):
        n = 96
        x = rng.standard_normal((n, D)).astype(np.float32)
        x[1] = 0.0                                  # constant row: value range 0 -> scale 1
        x[2] = 3.14159                              # constant, offset rounds away from the value
        x[3] = x[3] * 1e-7                          # scale underflows in float16 -> 1
        x[4] = x[4] * 1e4                           # large values
        x[5] = np.abs(x[5]) + 100.0                 # all positive, offset far from 0
        x[6] = -np.abs(x[6]) * 300.0                # all negative
        x[7, :] = 65000.0; x[7, 0] = -65000.0       # widest legal offset and a wide range
        x[8] = np.float32(1.0) + np.arange(D, dtype=np.float32) * np.float32(2 ** -12)  # ties at .5 after scaling
        x[9] = x[9].astype(np.float16).astype(np.float32)
        x[10, :] = 5e-8; x[10, 0] = -5e-8           # denormal float16 offset
        x[11:40] *= (10.0 ** rng.uniform(-4, 3, size=(29, 1))).astype(np.float32)
        q = Q.distributed_quantize_embeddings(x, D, "t", "QUint8RowwiseF16")
        out[f"x_{D}"], out[f"q_{D}"] = x, q
        out[f"dq_{D}"] = Q.dequantize_quint8_rowwise_f16(q, D)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_quant_vectors.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
