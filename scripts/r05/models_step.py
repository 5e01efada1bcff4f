#!/usr/bin/env python3
"""The three config-built train steps of bench.py's `secondary` (DeepFM-Criteo, multi_tower_din, MMoE + ZCH) alone:
`python scripts/r05/models_step.py [steps] [comma-separated keys]`."""
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torcheasyrec_amd import _build, _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None
_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)
bench.enable_tunable_gemm()
for k, r in bench.config_model_steps(dev, ws, steps=steps, only=only).items():
    print(json.dumps({"model": k, **{f: r.get(f) for f in ("ms_per_step", "host_queue_ms_per_step", "graph_ms_per_step", "graph_replays", "graph_error",
                                                           "loss", "error", "zch")}}), flush=True)
