#!/usr/bin/env python3
"""Stress of the backward plan (K6) alone, at the C-ABI level, under stream concurrency.

Zipf-clipped ids over ten mid-size tables at B = 65536 (hundreds of heavy buckets); every plan is
verified ON THE DEVICE: per table the sorted pairs are a permutation of the lookups, carry the right
row ids, keep equal rows adjacent and lookup positions ascending inside a row.  ~25 ms per iteration.

    python scripts/plan_stress.py <iters> <mode>[:knob=value[:knob=value]] [<mode>[:...] ...]

modes
    main        plan in the current stream, nothing else running
    side_idle   plan on a side stream, the main stream idle (synchronize before and after)
    side_noise  plan on a side stream while the main stream runs unrelated streaming kernels
    side_fwd    plan on a side stream while the main stream runs the pooled forward of the same ids
    side_apply  the module's async_plan shape: forward on main, plan on side, event, apply on main
                (no synchronize between plan and apply)
    main_apply  forward, plan, apply in one stream
    main_noise  plan in the main stream while a side stream runs unrelated streaming kernels
"""
import os
import sys
os.environ.setdefault("TZR_BWD_PLAN", "exact")  # (these scripts inspect the four-launch plan)
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402

ROWS = [12973, 11938, 39060, 17295, 7424, 20265, 7122, 2209, 3067956, 590152]


def make_batches(n, B, rows, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ids = np.stack([(np.minimum(rng.zipf(1.05, size=B).astype(np.int64) - 1, r - 1) * 2654435761 + 12345) % r
                        for r in rows])
        out.append(ids.reshape(-1))
    return out


def check_plan(ws, o8, values, F, B):
    """-> list of (table, what) problems; all on the device, a handful of syncs."""
    N = F * B
    pairs = ws[o8[0]:o8[0] + 8 * N].view(torch.int32).view(N, 2)
    k = pairs[:, 0].long().view(F, B)
    sp = pairs[:, 1].long().view(F, B)
    base = (torch.arange(F, device=ws.device) * B).view(F, 1)
    ar = torch.arange(B, device=ws.device).view(1, B)
    perm_ok = (torch.sort(sp, dim=1).values == base + ar).all(dim=1)
    key_ok = (values[sp.clamp(0, N - 1)] == k).all(dim=1)
    runs = (k[:, 1:] != k[:, :-1]).sum(dim=1) + 1
    ks = torch.sort(k, dim=1).values
    uniq = (ks[:, 1:] != ks[:, :-1]).sum(dim=1) + 1
    asc_ok = ((sp[:, 1:] > sp[:, :-1]) | (k[:, 1:] != k[:, :-1])).all(dim=1)
    good = perm_ok & key_ok & (runs == uniq) & asc_ok
    if bool(good.all()):
        return []
    bad = []
    for f in (~good).nonzero().flatten().tolist():
        what = (f"perm {bool(perm_ok[f])} keys {bool(key_ok[f])} runs {int(runs[f])} unique {int(uniq[f])} "
                f"ascending {bool(asc_ok[f])}")
        # which lookups never arrived, and which sorted positions hold something wrong
        rel = (sp[f] - f * B)
        inr = (rel >= 0) & (rel < B)
        cnt = torch.bincount(rel[inr], minlength=B)
        missing = (cnt == 0).nonzero().flatten()
        dup_src = (cnt > 1).nonzero().flatten()
        mrows = values[f * B + missing]
        # sorted positions whose content is not a first occurrence of a valid lookup with the right key
        seen_first = torch.zeros(B, dtype=torch.bool, device=ws.device)
        order = torch.argsort(rel.clamp(0, B - 1), stable=True)
        srt = rel.clamp(0, B - 1)[order]
        first = torch.ones(B, dtype=torch.bool, device=ws.device)
        first[1:] = srt[1:] != srt[:-1]
        seen_first[order[first]] = True
        wrong_pos = (~(inr & seen_first & (values[sp[f].clamp(0, N - 1)] == k[f]))).nonzero().flatten()
        what += (f"; missing lookups {missing.numel()} (first {missing[:6].tolist()}, rows {mrows[:6].tolist()}, "
                 f"distinct rows {torch.unique(mrows).numel()}, span {int(missing.min()) if missing.numel() else -1}.."
                 f"{int(missing.max()) if missing.numel() else -1}); duplicated lookups {dup_src.numel()}; wrong sorted positions "
                 f"{wrong_pos.numel()} (first {wrong_pos[:8].tolist()}, span {int(wrong_pos.min()) if wrong_pos.numel() else -1}.."
                 f"{int(wrong_pos.max()) if wrong_pos.numel() else -1}, mod16 histogram "
                 f"{torch.bincount(wrong_pos % 16, minlength=16).tolist() if wrong_pos.numel() else []})")
        bad.append((f, what))
    return bad


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    cases = sys.argv[2:] or ["main"]
    emu = bool(os.environ.get("STRESS_EMU"))
    if emu:  # logic dry run on the CPU lane emulator (mode main only)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        from emu.build_emu import build as build_emu
        _lib.use_library(build_emu())
    else:
        _lib.use_library(_build.build())
    L = _lib.lib()
    dev = torch.device("cpu") if emu else torch.device("cuda", 0)
    B = int(os.environ.get("STRESS_B", "4096" if emu else "65536"))
    rows = ROWS
    F = len(rows)
    keys = [f"c{i}" for i in range(F)]
    ebc = EmbeddingBagCollection([EmbeddingBagConfig(f"t{i}", 16, r, [k]) for i, (r, k) in enumerate(zip(rows, keys))],
                                 device=dev, optimizer=SparseOptimizerConfig(kind="sgd", lr=1e-3))
    host = make_batches(8, B, rows)
    batches = [KeyedJaggedTensor(keys, torch.from_numpy(v), torch.ones(F * B, dtype=torch.int32), uniform_length=1).to(dev)
               for v in host]
    g = torch.randn(B, 16 * F, device=dev) * 1e-3
    noise_a = torch.zeros(1 if emu else 32 << 20, device=dev)
    noise_b = torch.zeros(1 if emu else 32 << 20, device=dev)
    side = None if emu else torch.cuda.Stream(device=dev)
    import ctypes
    N = F * B
    o8 = (ctypes.c_int64 * 8)()
    assert L.tzr_pooled_bwd_plan_view(N, N, F, F, 16, o8) == 0
    sync = (lambda: None) if emu else torch.cuda.synchronize
    sync()
    total_bad = 0
    for case in cases:
        mode, *knobs = case.split(":")
        for name in (b"bwd_one_wg_heavy", b"bwd_ch"):
            L.tzr_tune(name, 0)
        L.tzr_tune(b"bwd_no_fuse_sort", 1)  # the plan is verified from ks[0]: every unit sorted by the sort launch
        for kv in knobs:
            name, v = kv.split("=")
            assert L.tzr_tune(name.encode(), int(v)) == 0, kv
        total_bad += run_case(mode, knobs, iters, emu, ebc, batches, g, noise_a, noise_b, side, dev, L, o8, F, B, rows, sync)
    return total_bad


def run_case(mode, knobs, iters, emu, ebc, batches, g, noise_a, noise_b, side, dev, L, o8, F, B, rows, sync):
    bad_iters = 0
    t0 = time.time()
    for it in range(iters):
        kjt = batches[it % len(batches)]
        cur = None if emu else torch.cuda.current_stream(dev)
        if mode == "main":
            ws = ebc.plan_backward(kjt)
        elif mode == "main_apply":
            ebc._launch_forward(kjt, ("__all__",))
            ws = ebc.plan_backward(kjt)
            ebc._launch_backward(kjt, ("__all__",), [g])
        elif mode == "main_noise":  # the plan in the main stream, unrelated kernels on the side stream
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(12):
                    torch.mul(noise_a, 1.0001, out=noise_b)
            ws = ebc.plan_backward(kjt)
        elif mode == "side_idle":
            torch.cuda.synchronize()
            with torch.cuda.stream(side):
                ws = ebc.plan_backward(kjt)
        elif mode == "side_noise":
            side.wait_stream(cur)
            for _ in range(6):
                torch.mul(noise_a, 1.0001, out=noise_b)
            with torch.cuda.stream(side):
                ws = ebc.plan_backward(kjt)
            for _ in range(6):
                torch.mul(noise_b, 1.0001, out=noise_a)
        elif mode == "side_fwd":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ws = ebc.plan_backward(kjt)
            ebc._launch_forward(kjt, ("__all__",))
        elif mode == "side_apply":
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ws = ebc.plan_backward(kjt)
                ev = torch.cuda.Event()
                ev.record()
            ebc._launch_forward(kjt, ("__all__",))
            (noise_b[: B * 16 * F].view(B, -1) * g).sum()
            cur.wait_event(ev)
            ebc._launch_backward(kjt, ("__all__",), [g])
        else:
            raise SystemExit(f"unknown mode {mode}")
        sync()
        probs = check_plan(ws, o8, kjt.values(), F, B)
        kjt._tzr_plan = None
        if probs:
            bad_iters += 1
            if bad_iters <= 8:
                for f, what in probs:
                    print(f"  iter {it} table {f} ({rows[f]} rows): {what}", flush=True)
    print(f"plan_stress mode {mode} knobs {knobs}: {bad_iters} bad of {iters} iterations, {time.time() - t0:.1f} s", flush=True)
    return bad_iters


if __name__ == "__main__":
    main()
