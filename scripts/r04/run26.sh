#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04al}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 60 python $R/scripts/r04/top_kernels_only.py > $O/plain.log 2>&1; echo "plain rc=$?"
i=0
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$i -o p --output-format csv -- python $R/scripts/r04/top_kernels_only.py > $O/pmc_$i.log 2>&1; echo "pmc $i rc=$?"
done
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'tzr_ia' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open("$O/pmc_top_kernels_summary.txt","w") as out:
    for k,d in acc.items():
        out.write(k+"\n")
        for c,v in sorted(d.items()):
            out.write(f"   {c:32s} {sum(v[1:])/max(1,len(v[1:])):16.1f}\n")
print(open("$O/pmc_top_kernels_summary.txt").read())
PY
