#!/usr/bin/env python3
"""K13 micro-benchmark: remap 26 x 65536 ids through a 100 M-row zero-collision-hash table (config-5
scale) and report ids/s and the HBM line rate it implies.  Per id the kernel
reads the id (8 B), writes the row (8 B) and touches ~1.5 cells of the open-addressing map: one
64-byte line for the key, one for the row (hit) -> ~144 B of compulsory line traffic per hit."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402
from torcheasyrec_amd.zch import ManagedCollisionModule, ZchConfig  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    Z = int(os.environ.get("ZCH_ROWS", 100_000_000))
    resident = Z // 2
    mod = ManagedCollisionModule(ZchConfig(Z, 5, "lfu"), dev)
    g = torch.Generator(device=dev).manual_seed(1)
    raw = torch.randint(0, 1 << 62, (resident,), device=dev, generator=g, dtype=torch.int64)
    raw = torch.unique(raw)
    mod.row_ids[:raw.numel()] = raw
    mod.rebuild()
    torch.cuda.synchronize()
    F, B = 26, 65536
    n = F * B
    hit = raw[torch.randint(0, raw.numel(), (n // 2,), device=dev, generator=g)]
    miss = torch.randint(0, 1 << 62, (n - n // 2,), device=dev, generator=g, dtype=torch.int64) | (1 << 62)
    vals = torch.cat([hit, miss])[torch.randperm(n, device=dev, generator=g)].contiguous()
    out = torch.empty_like(vals)
    s = mod.struct()
    d_mods = torch.frombuffer(bytearray(bytes(s)), dtype=torch.uint8).to(dev)
    km = torch.zeros(F, dtype=torch.int32, device=dev)
    cand = torch.empty(n, dtype=torch.int64, device=dev)
    L = _lib.lib()

    def run(profile):
        _lib.check(L.tzr_zch_remap(_lib.ptr(d_mods), _lib.ptr(km), F, _lib.ptr(vals), None, B, 1, n, 1, profile,
                                   _lib.ptr(out), _lib.ptr(cand), _lib.stream_ptr(dev)), "remap")

    res = {}
    for profile in (0, 1):
        for _ in range(3):
            run(profile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(profile)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        res["profile" if profile else "lookup"] = {"us": us, "ids_per_s": n / (us * 1e-6),
                                                   "line_GBps": n * 144 / (us * 1e-6) / 1e9}
    # one admission / eviction cycle with the table FULL (every candidate has to beat a resident)
    mod.row_ids[:Z - 1] = torch.arange(1, Z, device=dev, dtype=torch.int64) * 2 + (1 << 61)
    mod.counts[:Z - 1] = torch.randint(1, 50, (Z - 1,), device=dev, generator=g)
    mod.rebuild()
    cand_ids = torch.randint(0, 1 << 60, (n,), device=dev, generator=g, dtype=torch.int64)
    cand_ids = cand_ids[torch.randint(0, n, (4 * n,), device=dev, generator=g)]  # duplicates: counts up to ~10
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    changed = mod.update_and_evict(cand_ids, 7)
    e1.record()
    torch.cuda.synchronize()
    res["evict_cycle"] = {"ms": e0.elapsed_time(e1), "candidates": int(cand_ids.numel()), "rows_changed": int(changed.numel()),
                          "peak_GB": torch.cuda.max_memory_allocated() / 2**30}
    hits = int((out != Z - 1).sum().item())
    res.update({"zch_rows": Z, "resident": int(raw.numel()), "ids": n, "hits": hits, "candidates": int((cand != _lib.ZCH_EMPTY).sum().item())})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
