#!/usr/bin/env python3
"""A/B of the embedding path's three C-ABI calls (pooled forward, backward plan, backward apply)
under tzr_tune knob sets, on the real DLRM-Criteo tables, one process = one box:

    python scripts/emb_ab.py [--opt adagrad] [--B 65536,8192] [--dist uniform,zipf] "name=v,name=v" "..." ...

Each knob set (an empty string = defaults) is timed like bench.py times its `roofline` stages: HIP events
around every call on the launching stream, a GPU-side sleep first so the host is not in the gaps."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from torcheasyrec_amd import _build, _lib  # noqa: E402
from torcheasyrec_amd.criteo import CRITEO_ROWS, SPARSE_KEYS, algorithmic_bytes, criteo_tables, synthetic_batch  # noqa: E402
from torcheasyrec_amd.embedding import EmbeddingBagCollection, SparseOptimizerConfig  # noqa: E402

KNOBS = [b"bwd_apply_waves", b"bwd_apply_fast", b"bwd_no_fuse_sort", b"bwd_scan_slices", b"fwd_tile_b", b"fwd_variant", b"bwd_ch", b"bwd_one_wg_heavy", b"bwd_force_prep"]


class Timers:
    def __init__(self):
        self.pairs = {}

    def start(self, name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.pairs.setdefault(name, []).append((e0, e1))
        return e1

    def us(self, name):
        ps = self.pairs.get(name, [])
        v = sorted(a.elapsed_time(b) * 1e3 for a, b in ps)
        return (float(np.mean(v)), v[len(v) // 2], v[0]) if v else (0.0, 0.0, 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", default="adagrad")
    ap.add_argument("--B", default="65536")
    ap.add_argument("--dist", default="uniform")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--layout", default="interleaved")
    ap.add_argument("--lib", default=None, help="another build of the library to time (same-box A/B against an older tree)")
    ap.add_argument("--grad-stride0", action="store_true",
                    help="experiment: the apply reads every lookup's gradient from ONE row (sample stride 0): what the apply costs without "
                         "its 1.7 M random gradient-row requests (results are meaningless)")
    ap.add_argument("--rows-cap", type=int, default=0, help="experiment: every table capped at this many rows (address-translation reach)")
    ap.add_argument("--plan", default="auto", help="EmbeddingBagCollection.plan_mode: auto | cells | exact (comma list: each is timed)")
    ap.add_argument("sets", nargs="*", default=[""])
    args = ap.parse_args()
    _lib.use_library(args.lib or _build.build())
    L = _lib.lib()
    if args.grad_stride0:
        orig_apply = L.tzr_pooled_bwd_apply

        def apply0(*a):
            for i in range(a[13]):
                a[12][i].stride = 16  # (rows overlap: the whole gradient "buffer" is 4 MB, cache-resident, spread over every channel)
            return orig_apply(*a)

        L.tzr_pooled_bwd_apply = apply0
    dev = torch.device("cuda", 0)
    ROWS = [min(r, args.rows_cap) for r in CRITEO_ROWS] if args.rows_cap else list(CRITEO_ROWS)
    ebc = EmbeddingBagCollection(criteo_tables(ROWS), device=dev,
                                 optimizer=SparseOptimizerConfig(kind=args.opt, lr=1e-3), groups={"sparse": SPARSE_KEYS},
                                 row_layout=args.layout)
    for B in [int(x) for x in args.B.split(",")]:
        for dist in args.dist.split(","):
            host = [synthetic_batch(s, B, ROWS, dist=dist)[1] for s in range(4)]
            ab = [algorithmic_bytes(k.values().numpy(), B, ROWS, optimizer=args.opt) for k in host]
            nbytes = float(np.mean([a["fwd"] + a["bwd"] for a in ab]))
            batches = [k.to(dev) for k in host]
            g = torch.randn(B, 416, device=dev) * 1e-3
            for spec in [(sp, pm) for pm in args.plan.split(",") for sp in args.sets]:
                spec, ebc.plan_mode = spec
                for k in KNOBS:
                    L.tzr_tune(k, 0)
                for kv in [x for x in spec.split(",") if x]:
                    name, v = kv.split("=")
                    assert L.tzr_tune(name.encode(), int(v)) == 0, kv
                for i in range(3):
                    kjt = batches[i % 4]
                    ebc._launch_forward(kjt, ("sparse",))
                    ebc.plan_backward(kjt, ("sparse",))
                    ebc._launch_backward(kjt, ("sparse",), [g])
                torch.cuda.synchronize()
                tm = Timers()
                torch.cuda._sleep(int(2.0e7))
                ebc._timers = tm
                for i in range(args.iters):
                    kjt = batches[i % 4]
                    ebc._launch_forward(kjt, ("sparse",))
                    ebc.plan_backward(kjt, ("sparse",))
                    ebc._launch_backward(kjt, ("sparse",), [g])
                torch.cuda.synchronize()
                ebc._timers = None
                f, p, a = tm.us("fwd"), tm.us("plan"), tm.us("apply")
                tot = f[0] + p[0] + a[0]
                print(f"B {B:6d} {dist:8s} {args.opt + '/' + args.layout[:5]:22s} [{(spec or 'defaults') + ' plan=' + ebc.plan_mode:40s}] fwd {f[0]:6.1f} (med {f[1]:6.1f} min {f[2]:6.1f})  "
                      f"plan {p[0]:6.1f} ({p[1]:6.1f} {p[2]:6.1f})  apply {a[0]:6.1f} ({a[1]:6.1f} {a[2]:6.1f})  "
                      f"sum {tot:6.1f} us  frac {nbytes / (tot * 1e-6) / 8e12:.3f}", flush=True)


if __name__ == "__main__":
    main()
