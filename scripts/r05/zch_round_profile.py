#!/usr/bin/env python3
"""Where an admission / eviction round of a mostly EMPTY 200 M-row zero-collision-hash table spends its time (bench.py's
secondary.mmoe_zch_b8192.zch.round_ms): update_and_evict under cProfile, with a device synchronize after every torch call of its
body (sections timed by hand below).  `python scripts/r05/zch_round_profile.py [rows] [candidates]`"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.zch import ManagedCollisionModule, ZchConfig  # noqa: E402

_lib.use_library(_build.build())
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_949_696
for rnd in range(3):
    m = ManagedCollisionModule(ZchConfig(rows, 1000, "lfu", 1.0), dev) if rnd == 0 else m
    g = torch.Generator(device=dev).manual_seed(rnd)
    # Zipf-like raw ids: few thousand distinct frequent ones + a long tail (what the bench's user ids look like)
    u = torch.empty(n, device=dev).exponential_(1.0, generator=g)
    cand = ((u * 1e4).long() ** 2 + rnd * 7919) * 2654435761 % (1 << 62)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    changed = m.update_and_evict(cand, 1000 * (rnd + 1))
    torch.cuda.synchronize()
    pr.disable()
    print(f"round {rnd}: {1e3 * (time.perf_counter() - t0):.1f} ms, {cand.numel()} candidates, {changed.numel()} rows changed, "
          f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
