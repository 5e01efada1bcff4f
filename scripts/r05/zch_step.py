#!/usr/bin/env python3
"""MMoE + 200 M-row zero-collision hash train step (bench.py secondary.mmoe_zch_b8192) alone: `python scripts/r05/zch_step.py [steps]`."""
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torcheasyrec_amd import _build, _lib  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)
bench.enable_tunable_gemm()
r = bench.config_model_steps(dev, ws, steps=steps, only={"mmoe_zch_b8192"})["mmoe_zch_b8192"]
print(json.dumps({k: r.get(k) for k in ("ms_per_step", "host_queue_ms_per_step", "graph_ms_per_step", "graph_error", "loss", "error", "zch")}), flush=True)
