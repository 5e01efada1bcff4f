#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04ad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 60 python $R/scripts/r04/wgrad_only.py > $O/plain.log 2>&1; echo "plain rc=$?"
for c in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" ; do
  tag=$(echo $c | tr ' ' '_')
  timeout 90 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o p --output-format csv -- python $R/scripts/r04/wgrad_only.py > $O/pmc_$tag.log 2>&1; echo "pmc $c rc=$?"
  f=$(find $O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep "wgrad_kernel" $f | awk -F, '{print $(NF-1), $NF}' | sort | uniq -c | head -8
done
