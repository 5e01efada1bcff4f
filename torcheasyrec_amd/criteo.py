"""DLRM-Criteo workload definition (tables of /root/reference/examples/dlrm_criteo.config:124-329)
and the synthetic batch generator of SURVEY.md section 8(d) / BASELINE.md section 2."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .embedding import EmbeddingBagConfig
from .sparse import KeyedJaggedTensor

# num_buckets of cat_0 .. cat_25, in config order
CRITEO_ROWS: List[int] = [
    40000000, 39060, 17295, 7424, 20265, 3, 7122, 1543, 63, 40000000, 3067956, 405282, 10, 2209,
    11938, 155, 4, 976, 14, 40000000, 40000000, 40000000, 590152, 12973, 108, 36,
]
NUM_DENSE = 13
EMBEDDING_DIM = 16
SPARSE_KEYS = [f"cat_{i}" for i in range(26)]
DENSE_KEYS = [f"int_{i}" for i in range(NUM_DENSE)]
SEED0 = 20260925


def criteo_tables(rows: Optional[List[int]] = None, dim: int = EMBEDDING_DIM, suffix: str = "_emb",
                  init: str = "default") -> List[EmbeddingBagConfig]:
    """One table per sparse feature, named ``{feature}_emb`` (tzrec/features/feature.py:615).

    init="default": module default (uniform +-sqrt(1/rows) drawn on the table's device);
    init="seeded":  the same distribution from torch.Generator().manual_seed(1000 + t) on the host
                    (used where the oracle must see identical weights)."""
    rows = rows or CRITEO_ROWS
    out = []
    for t, (k, r) in enumerate(zip(SPARSE_KEYS, rows)):
        fn = None
        if init == "seeded":
            def fn(w, t=t, r=r):  # noqa: E306
                g = torch.Generator().manual_seed(1000 + t)
                a = math.sqrt(1.0 / r)
                w.copy_((torch.rand(w.shape, generator=g) * 2 - 1) * a)
        out.append(EmbeddingBagConfig(f"{k}{suffix}", dim, r, [k], "sum", init_fn=fn))
    return out


def synthetic_batch(
    step: int, B: int, rows: Optional[List[int]] = None, dist: str = "uniform"
) -> Tuple[torch.Tensor, KeyedJaggedTensor, torch.Tensor]:
    """(dense [B,13] f32 = log(x+3), sparse KJT with one id per (sample, feature), label int64[B]).

    ids: numpy PCG64(seed = 20260925 + 131*step + t), uniform in [0, rows_t) or Zipf(1.05)
    rank-permuted and clipped to rows_t; dense x ~ randint(0, 1000); label ~ Bernoulli(0.25)."""
    rows = rows or CRITEO_ROWS
    vals = []
    for t, r in enumerate(rows):
        g = np.random.Generator(np.random.PCG64(SEED0 + 131 * step + t))
        if dist == "uniform":
            ids = g.integers(0, r, size=B, dtype=np.int64)
        elif dist == "zipf":
            z = g.zipf(1.05, size=B).astype(np.int64) - 1
            # rank -> row through a fixed multiplicative permutation so hot rows are scattered
            ids = (np.minimum(z, r - 1) * 2654435761 + 12345) % r
        else:
            raise ValueError(dist)
        vals.append(ids)
    g = np.random.Generator(np.random.PCG64(SEED0 + 131 * step + 1000))
    dense = np.log(g.integers(0, 1000, size=(B, NUM_DENSE)).astype(np.float32) + 3.0).astype(np.float32)
    label = (g.random(B) < 0.25).astype(np.int64)
    kjt = KeyedJaggedTensor(
        SPARSE_KEYS[: len(rows)],
        torch.from_numpy(np.concatenate(vals)),
        torch.ones(len(rows) * B, dtype=torch.int32),
        uniform_length=1,
    )
    return torch.from_numpy(dense), kjt, torch.from_numpy(label)


def algorithmic_bytes(kjt_values: np.ndarray, B: int, rows: List[int], dim: int = EMBEDDING_DIM,
                      optimizer: str = "adagrad") -> Dict[str, float]:
    """Compulsory HBM bytes of the pooled forward / backward for one batch:
    fwd = 8N + 4FB + 4D*U + 4*sumD*B; bwd adagrad = 4*sumD*B + 8N + 16D*U;
    bwd rowwise = 4*sumD*B + 8N + (8D+8)*U, with U = distinct (table,row) pairs in the batch."""
    F = len(rows)
    N = len(kjt_values)
    U = 0
    for t in range(F):
        U += len(np.unique(kjt_values[t * B:(t + 1) * B]))
    sumD = F * dim
    fwd = 8 * N + 4 * F * B + 4 * dim * U + 4 * sumD * B
    if optimizer == "adagrad":
        bwd = 4 * sumD * B + 8 * N + 16 * dim * U
    elif optimizer == "rowwise_adagrad":
        bwd = 4 * sumD * B + 8 * N + (8 * dim + 8) * U
    else:
        bwd = 4 * sumD * B + 8 * N + 8 * dim * U
    return {"N": N, "U": U, "fwd": float(fwd), "bwd": float(bwd)}
