#!/usr/bin/env python3
"""multi_tower_din train step (bench.py secondary.din_taobao_b8192) alone, DIN towers on the jagged positions and on the
padded tensors in one process: `python scripts/r05/din_step.py [steps] [jagged|padded|both]`."""
import json
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torcheasyrec_amd import _build, _lib, rank_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
which = sys.argv[2] if len(sys.argv) > 2 else "both"
_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)
bench.enable_tunable_gemm()
for mode in (["jagged", "padded"] if which == "both" else [which]):
    rank_model.JAGGED_DIN = mode == "jagged"
    r = bench.config_model_steps(dev, ws, steps=steps, only={"din_taobao_b8192"})["din_taobao_b8192"]
    print(json.dumps({"din_towers": mode, **{k: r.get(k) for k in ("ms_per_step", "host_queue_ms_per_step", "graph_ms_per_step", "graph_error", "loss", "error", "ids_per_step")}}), flush=True)
