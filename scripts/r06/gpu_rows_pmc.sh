#!/bin/bash
# gpurun -- bash scripts/r06/gpu_rows_pmc.sh <tag>: SQ counters of the tall-input Linear kernels (one pass per counter set)
tag=${1:-r06p}; out=gpurun_out/$tag; mkdir -p $out; R=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*Counter_Name|Name *:" | sed 's/.*:\s*//' | sort -u | tr '\n' ' ' > $R/$out/counters.txt
run() {  # name, counters..., then -- args
  name=$1; shift; cs=(); while [ "$1" != "--" ]; do cs+=("$1"); shift; done; shift
  for k in "fwd 96 256" "fwd 256 64" "wg 96 256" "dx 256 96"; do
    kk=$(echo $k | tr ' ' '_')
    timeout 200 rocprofv3 --kernel-trace --pmc "${cs[@]}" -d $R/$out/p_${name}_$kk -o p --output-format csv -- python $R/scripts/r06/rows_gemm_one.py $k > /dev/null 2>&1
    echo "$name $k rc=$?"
  done
}
run a SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --
run c SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --
run d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --
run e SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --
cd $R
python - <<'PY' $out
import csv, glob, os, sys, collections
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/p_*")):
    key = os.path.basename(d)[4:]  # kind_K_H
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "gemm_rows" not in kn and "gemm_tn_kernel" not in kn:
                continue
            c, v = row["Counter_Name"], float(row["Counter_Value"])
            res[key].setdefault(c, []).append(v)
with open(out + "/pmc_rows_summary.txt", "w") as fo:
    for key, cs in res.items():
        fo.write(key + "\n")
        for c, vs in sorted(cs.items()):
            fo.write("   %-28s mean per launch %.4g  (%d launches)\n" % (c, sum(vs) / len(vs), len(vs)))
print(open(out + "/pmc_rows_summary.txt").read())
PY
rm -rf $out/p_*
