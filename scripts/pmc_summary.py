#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in
separate runs, MI355X_MICROARCH.md: they do not fit one pass).

    pmc_summary.py <fetch_dir> <write_dir> <out.json> [n_lookups]

Counter values are KiB per dispatch.  Corrections applied, as calibrated with scripts/probe_hbm.hip on
this chip (profiles/r01c/probe_hbm_access_patterns.txt): WRITE_SIZE is exact; FETCH_SIZE counts 64 B
per fabric read request, i.e. exact for 64-byte row gathers and HALF of a wide streaming read -- so
for the pooled forward the streamed id array (8 B x n_lookups) is added back once more at 1/2."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "tzr_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]) * 1024.0)
    # skip the first (cold) dispatch of every kernel when there are several
    return {k: sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0] for k, v in acc.items()}


def main(fetch_dir, write_dir, out, n_lookups=26 * 65536):
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    ks = {}
    for k in sorted(set(fe) | set(wr)):
        ks[k] = {"FETCH_SIZE": fe.get(k), "WRITE_SIZE": wr.get(k)}
    # the forward runs as tzr_pooled_fwd_u1_kernel (ids staged in LDS) at B >= 32768 one-id bags, else tzr_pooled_fwd_kernel
    fwd_name = "tzr_pooled_fwd_u1_kernel" if "tzr_pooled_fwd_u1_kernel" in ks else "tzr_pooled_fwd_kernel"
    # ... and, in a training step whose backward takes the cells plan, as the launch that carries the plan's workgroups behind its own
    fused = "tzr_pooled_fwd_u1_cells_plan_kernel"
    if fused in ks:
        fwd_name = fused
    if fwd_name in ks:
        e = ks[fwd_name]
        e["traffic_corrected"] = e["FETCH_SIZE"] + 0.5 * 8 * int(n_lookups) + e["WRITE_SIZE"]
    # the north-star aggregate: the six launches of the pooled embedding forward + backward
    red_name = next((k for k in ("tzr_bwd_reduce_fast_adagrad_kernel", "tzr_bwd_reduce_fast_rowwise_kernel", "tzr_bwd_reduce_fast_sgd_kernel",
                                 "tzr_bwd_reduce_w7_kernel", "tzr_bwd_reduce_w8_kernel", "tzr_bwd_reduce_kernel") if k in ks),
                    "tzr_bwd_reduce_kernel")  # the apply's instantiation for the run's optimizer (round 5: the fast tile loop)
    six = [fwd_name, "tzr_bwd_hist_kernel", "tzr_bwd_scan_kernel", "tzr_bwd_scatter_kernel",
           "tzr_bwd_sort_kernel", red_name]
    # round 6: batches of one id per bag take the one-launch index plan (csrc/pooled_bwd_cells.hip): THREE launches in all
    cells_apply = next((k for k in ks if k.startswith("tzr_bwd_cells_apply_")), None)
    if fwd_name == fused and cells_apply:
        six = [fused, cells_apply]  # TWO launches
    elif "tzr_bwd_cells_partition_kernel" in ks and cells_apply:
        six = [fwd_name, "tzr_bwd_cells_partition_kernel", cells_apply]
    if all(k in ks for k in six):
        agg = 0.0
        for k in six:
            e = ks[k]
            agg += e.get("traffic_corrected") or ((e["FETCH_SIZE"] or 0.0) + (e["WRITE_SIZE"] or 0.0))
        ks["__embedding_fwd_bwd__"] = {"traffic_corrected": agg, "kernels": six,
                                       "note": "forward corrected as above; the others at FETCH_SIZE + WRITE_SIZE as counted "
                                               "(64-B row gathers are exact; the 8-B key/source streams of the plan are uncalibrated)"}
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from torcheasyrec_amd import _build  # the digest of the kernel sources these counters were measured on

    json.dump({"lib_digest": _build._digest(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --steps 4 "
                         "--warmup 2 --no-cpu-baseline --no-graph; B=65536 uniform ids, adagrad interleaved",
               "units": "bytes per launch (mean over dispatches after the first); see scripts/pmc_summary.py for the corrections",
               "kernels": ks}, open(out, "w"), indent=1)
    print(json.dumps(ks.get(fwd_name)))


if __name__ == "__main__":
    main(*sys.argv[1:])
