#!/bin/bash
# SQ issue / wait counters of the embedding path's kernels (one pass: 8 SQ slots): is a kernel bound by instruction issue or by waiting?
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05b}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O/pmc_sq -o p --output-format csv -- python $R/scripts/plan_trace.py ${2:-uniform} ${3:-65536} ${4:-adagrad} > $O/pmc_sq.log 2>&1; echo "pmc rc=$?"
cd $R
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/pmc_sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][:44]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            cnt[k] += 1
out = open(O + "/sq_counters.txt", "w")
hdr = "%-46s %6s %10s %8s %8s %8s %8s %9s %9s" % ("kernel", "calls", "wave_cyc/w", "wait%", "stall%", "active%", "valu%", "valu/wave", "salu/wave")
print(hdr); out.write(hdr + "\n")
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = max(c.get("SQ_WAVES", 1), 1)
    wc = max(c.get("SQ_WAVE_CYCLES", 1), 1)
    line = "%-46s %6d %10.0f %8.1f %8.1f %8.1f %8.1f %9.0f %9.0f" % (
        k, cnt[k], 4 * wc / w, 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc,
        100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w)
    print(line); out.write(line + "\n")
PY
rm -rf $O/pmc_sq
