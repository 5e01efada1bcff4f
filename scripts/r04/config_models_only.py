#!/usr/bin/env python3
"""bench.py's config-model step lines alone (DeepFM / multi_tower_din / MMoE + ZCH at batch 8192)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

bench.enable_tunable_gemm()
dev = torch.device("cuda", 0)
ws = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(ws)  # (as bench.py's main does: the whole run on one side stream)
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None  # e.g. din_taobao_b8192
res = bench.config_model_steps(dev, ws, steps=20, only=only)
for k, v in res.items():
    print(k, json.dumps({kk: v.get(kk) for kk in ("ms_per_step", "graph_ms_per_step", "host_queue_ms_per_step", "graph_error", "zch")}))
