#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage):
the offline check every change of a kernel compiled for a fixed number of waves per SIMD goes through (no GPU needed).
usage: python scripts/r05/kres.py pooled_bwd_apply.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "torcheasyrec_amd", "csrc")
f, extra = sys.argv[1], sys.argv[2:]
p = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-I", ".", "-c", f, "-o", "/tmp/kres.o",
                    "-Rpass-analysis=kernel-resource-usage", *extra], cwd=csrc, capture_output=True, text=True)
cur = {}
for l in p.stderr.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", l)
    if not m:
        if "error" in l or "warning:" in l:
            print(l.rstrip())
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            n = re.sub(r"^_Z\d+", "", cur["name"])[:52]
            print("%-54s VGPR %4s AGPR %3s SGPR %4s scratch %5s occ %2s LDS %s" % (
                n, cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize [bytes/lane]"),
                cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
sys.exit(p.returncode)
