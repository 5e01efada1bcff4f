#!/usr/bin/env python3
"""MFMA activity of the dot-interaction kernels (stand-alone and fused with the first top-MLP layer) and of the small-MLP
kernels from one rocprofv3 PMC pass.

    pmc_mfma_summary.py <pmc_dir> <out.json>

Counters (per dispatch, summed over the chip by rocprofv3): SQ_INSTS_VALU_MFMA_MOPS_F32 (MFMA
operations in units of 512 flop), SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe was busy, summed over
the SIMDs), SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE (cycles the kernel ran).  Reported per kernel: flop, the
share of SIMD-cycles with a busy MFMA pipe (utilisation against the chip: 256 CUs x 4 SIMDs), and
the achieved MFMA rate against the fp32 dense peak (157.3 TFLOP/s, MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

PEAK_F32_MFMA = 157.3e12
SIMDS = 256 * 4


def main(d, out):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "interaction" not in k and "tzr_ia_" not in k and "tzr_mlp" not in k:  # the MFMA kernels of the dense half
            continue
        name = k.split("(")[0].replace("void ", "").split("<")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for name, cs in acc.items():
        m = {c: sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0] for c, v in cs.items()}
        e = {"counters": m}
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
            e["mfma_flop"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
        if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0  # reported summed over the 8 XCDs (a 58 us kernel reads 1.16 M = 8 x 145 k)
            e["kernel_cycles"] = cyc
            e["mfma_busy_share_of_simd_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS)
            if "mfma_flop" in e:
                e["mfma_tflops_while_running"] = e["mfma_flop"] / (cyc / 2.4e9) / 1e12
                e["frac_of_f32_mfma_peak"] = e["mfma_tflops_while_running"] * 1e12 / PEAK_F32_MFMA
        res[name] = e
    json.dump({"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
                         "GRBM_GUI_ACTIVE -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e --no-graph",
               "peak_f32_mfma_flops": PEAK_F32_MFMA, "kernels": res}, open(out, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in res.items()}))


if __name__ == "__main__":
    main(*sys.argv[1:])
