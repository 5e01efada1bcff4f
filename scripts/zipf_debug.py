#!/usr/bin/env python3
"""Stress of the backward plan + apply under heavy skew: eight mid-size tables, Zipf-clipped ids at
B = 65536 (hundreds of heavy buckets, stitch groups sharing record lines), SGD so every row has a
closed form; every iteration is checked on the device against index_add in fp64.

    python scripts/zipf_debug.py [iterations] [bwd_debug ...]"""
import os
import sys
os.environ.setdefault("TZR_BWD_PLAN", "exact")  # (these scripts inspect the four-launch plan)

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _lib  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, EmbeddingBagConfig, SparseOptimizerConfig  # noqa: E402
from torcheasyrec_amd.sparse import KeyedJaggedTensor  # noqa: E402


def run(iters, dbg, B=65536, seed=0, variant=0):
    dev = torch.device("cuda", 0)
    rows = [12973, 11938, 39060, 17295, 7424, 20265, 7122, 2209, 3067956, 590152]
    keys = [f"c{i}" for i in range(len(rows))]
    lr = 0.5
    ebc = EmbeddingBagCollection([EmbeddingBagConfig(f"t{i}", 16, r, [k]) for i, (r, k) in enumerate(zip(rows, keys))],
                                 device=dev, optimizer=SparseOptimizerConfig(kind="sgd", lr=lr))
    rng = np.random.default_rng(seed)
    _lib.lib().tzr_tune(b"bwd_one_wg_heavy", int(os.environ.get("TZR_ONE_WG_HEAVY", "0")))
    _lib.lib().tzr_tune(b"bwd_no_fuse_sort", 1)  # the plan is verified from ks[0]: every unit sorted by the sort launch
    bad = 0
    import time
    t0 = time.time()
    for it in range(iters):
        if it < 3 or it % 10 == 0:
            torch.cuda.synchronize()
            print(f"debug {dbg} variant {variant} iter {it} t={time.time() - t0:.1f}s", flush=True)
        ids = np.stack([(np.minimum(rng.zipf(1.05, size=B).astype(np.int64) - 1, r - 1) * 2654435761 + 12345) % r for r in rows])
        kjt = KeyedJaggedTensor(keys, torch.from_numpy(ids.reshape(-1)), torch.ones(len(rows) * B, dtype=torch.int32),
                                uniform_length=1).to(dev)
        before = [w.detach().clone() for w in ebc.table_weights().values()]
        g = torch.randn(B, 16 * len(rows), device=dev)
        idt = kjt.values().view(len(rows), B)
        # variant 0: the module's default -- plan on a side stream during the forward (async_plan)
        #         1: the same + a device synchronize before the backward
        #         2: plan on the main stream inside the backward
        #         3: plan on the main stream, synchronize, then backward
        ebc.async_plan = variant in (0, 1)
        out = ebc(kjt).values()
        if variant == 3:
            ebc.plan_backward(kjt)
        if variant in (1, 3):
            torch.cuda.synchronize()
        plan = getattr(kjt, "_tzr_plan", None)
        ws = plan[2] if plan is not None else None
        (out * g).sum().backward()
        torch.cuda.synchronize()
        if ws is not None:
            # the plan on its own: every table's sorted pairs must be a permutation of its lookups with equal
            # rows adjacent and ascending lookup positions inside a row (tzr_pooled_bwd_plan_view)
            N = len(rows) * B
            import ctypes
            o8 = (ctypes.c_int64 * 8)()
            assert _lib.lib().tzr_pooled_bwd_plan_view(N, N, len(rows), len(rows), 16, o8) == 0
            pairs = ws[o8[0]:o8[0] + 8 * N].view(torch.int32).view(N, 2)
            for f in range(len(rows)):
                k, sp = pairs[f * B:(f + 1) * B, 0].long(), pairs[f * B:(f + 1) * B, 1].long()
                perm_ok = bool((torch.sort(sp).values == torch.arange(f * B, (f + 1) * B, device=dev)).all())
                key_ok = perm_ok and bool((kjt.values()[sp.clamp(0, N - 1)] == k).all())
                runs = int((k[1:] != k[:-1]).sum()) + 1
                uniq = int(torch.unique(k).numel())
                asc_ok = bool(((sp[1:] > sp[:-1]) | (k[1:] != k[:-1])).all())
                if not (perm_ok and key_ok and runs == uniq and asc_ok):
                    bad += 1
                    print(f"debug {dbg} variant {variant} iter {it} PLAN table {f} ({rows[f]} rows): perm {perm_ok} keys {key_ok} "
                          f"runs {runs} unique {uniq} ascending {asc_ok}", flush=True)
        for f, w in enumerate(ebc.table_weights().values()):
            # per-row gradient sums without atomics (index_add_ on 40k duplicates of one row takes seconds in
            # fp64): sort the lookups by row, prefix-sum in fp64, difference at the run ends
            order = torch.argsort(idt[f], stable=True)
            sid = idt[f][order]
            gf = g[:, f * 16:(f + 1) * 16].double()[order]
            cs, ca = gf.cumsum(0), gf.abs().cumsum(0)
            end = torch.ones(B, dtype=torch.bool, device=dev)
            end[:-1] = sid[1:] != sid[:-1]
            e_idx = end.nonzero().squeeze(1)
            seg = cs[e_idx].clone()
            sega = ca[e_idx].clone()
            seg[1:] -= cs[e_idx[:-1]]
            sega[1:] -= ca[e_idx[:-1]]
            uniq = sid[e_idx]
            exp = before[f][uniq].double() - lr * seg
            err = (w.detach()[uniq].double() - exp).abs()
            bound = 1e-6 + 4 * 1.2e-7 * lr * sega + 1e-6 * exp.abs()
            moved = int((w.detach() != before[f]).any(dim=1).sum())
            if not bool((err <= bound).all()) or moved > uniq.numel():
                bad += 1
                r = int((err - bound).max(dim=1).values.argmax())
                print(f"debug {dbg} iter {it} table {f} ({rows[f]} rows): row {int(uniq[r])} err {float(err[r].max()):.3g} "
                      f"lookups of that row {int((idt[f] == uniq[r]).sum())}; rows moved {moved} vs touched {uniq.numel()}", flush=True)
    return bad


if __name__ == "__main__":
    _lib.use_native()
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    for a in sys.argv[2:] or ["0"]:  # "<bwd_debug>[:<variant>]"
        dbg, variant = (int(x) for x in (a.split(":") + ["0"])[:2])
        print("debug", dbg, "variant", variant, "failures", run(iters, dbg, variant=variant), "in", iters, "iterations", flush=True)
