#!/usr/bin/env python3
"""Admission / eviction round of one zero-collision-hash table at scale: the radix selection of csrc/zch_evict.hip
(ManagedCollisionModule._select_kept) against the three stable sorts it replaced, and the whole update_and_evict round.

    python scripts/bench_zch_evict.py [rows=200000000] [candidates=1703936] [policy=lfu]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.zch import EMPTY, ManagedCollisionModule, ZchConfig  # noqa: E402


def three_sorts(m, new_ids, new_cnt, cur_iter):
    """what zch.py did before: cat residents + candidates, three stable sorts, first Z - 1"""
    Z = m.cfg.zch_size
    res_rows = torch.nonzero(m.row_ids[:Z - 1] != EMPTY).squeeze(1)
    res_ids = m.row_ids[res_rows]
    s_res = m.counts[res_rows].double()
    ids = torch.cat([res_ids, new_ids])
    sc = torch.cat([s_res, new_cnt.double()])
    is_new = torch.cat([torch.zeros_like(res_ids), torch.ones_like(new_ids)])
    o = torch.sort(ids, stable=True).indices
    o = o[torch.sort(is_new[o], stable=True).indices]
    o = o[torch.sort(sc[o], descending=True, stable=True).indices]
    kept = o[:Z - 1]
    row_kept = torch.zeros(Z - 1, dtype=torch.uint8, device=ids.device)
    row_kept[res_rows[kept[kept < res_ids.numel()]]] = 1
    new_kept = torch.zeros(new_ids.numel(), dtype=torch.uint8, device=ids.device)
    new_kept[kept[kept >= res_ids.numel()] - res_ids.numel()] = 1
    return row_kept, new_kept


def main():
    cpu = os.environ.get("TZR_DRY_RUN_EMU")  # dry run of this script's logic on the lane emulator (no timings worth reading)
    if cpu:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from emu.build_emu import build

        _lib.use_library(build())
    else:
        _lib.use_native()
    dev = torch.device("cpu") if cpu else torch.device("cuda", 0)
    sync = (lambda: None) if cpu else torch.cuda.synchronize
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 26 * 65536
    policy = sys.argv[3] if len(sys.argv) > 3 else "lfu"
    m = ManagedCollisionModule(ZchConfig(rows + 1, 5, policy, 1.0), dev)
    g = torch.Generator(device=dev).manual_seed(1)
    # a full table: row r holds raw id 3r + 1; counts geometric (most rows seen once or twice), ages spread
    m.row_ids[:rows] = torch.arange(rows, device=dev, dtype=torch.int64) * 3 + 1
    m.counts[:rows] = torch.empty(rows, device=dev).exponential_(0.7, generator=g).long() + 1
    m.last_iter[:rows] = 1000 - torch.randint(0, 200, (rows,), device=dev, generator=g)
    new_ids = torch.sort(torch.randperm(rows, device=dev, generator=g)[:n] * 3 + 2).values  # never resident
    n = new_ids.numel()
    new_cnt = torch.empty(n, device=dev).exponential_(0.5, generator=g).long() + 1
    cur = 1000

    def timed(fn, reps):
        fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        sync()
        return (time.perf_counter() - t0) / reps * 1e3, out

    t_sel, (rk, nk) = timed(lambda: m._select_kept(new_ids, new_cnt, cur), 3)
    print(f"rows {rows}  candidates {n}  policy {policy}: radix selection {t_sel:.2f} ms "
          f"({int(nk.sum())} candidates admitted, {rows - int(rk.sum())} residents evicted)", flush=True)
    if policy == "lfu" and rows <= 250_000_000:
        t_sort, (rk2, nk2) = timed(lambda: three_sorts(m, new_ids, new_cnt, cur), 1)
        same = bool(torch.equal(rk, rk2) and torch.equal(nk, nk2))
        print(f"three stable sorts over residents + candidates: {t_sort:.1f} ms (same answer: {same}); "
              f"peak memory {0 if cpu else torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    cand = new_ids.repeat_interleave(new_cnt.clamp(max=3))
    sync()
    t0 = time.perf_counter()
    changed = m.update_and_evict(cand, cur)
    sync()
    print(f"whole round, first call (unique + selection + row hand-out + map update in place; cold allocator): "
          f"{(time.perf_counter() - t0) * 1e3:.1f} ms, {changed.numel()} rows changed owner", flush=True)
    # a second round with other candidates (ids 3r + 3: never resident, never seen): the steady-state cost
    cand2 = (new_ids + 1).repeat_interleave(new_cnt.clamp(max=3))
    sync()
    t0 = time.perf_counter()
    changed = m.update_and_evict(cand2, cur + 5)
    sync()
    print(f"whole round, second call: {(time.perf_counter() - t0) * 1e3:.1f} ms, {changed.numel()} rows changed owner", flush=True)


if __name__ == "__main__":
    main()
