#!/bin/bash
# the fused hidden-layer backward of the DIN attention MLP (tzr_linear_bwd_relu, tzr_head_bwd_relu): parity tests on the GPU, the
# kernel against the GEMM + mask pair it replaces at the Taobao shape, the step, and its kernel table
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05x}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dense_glue.py tests/test_sequence_parity.py -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/gpu_tests.log | cut -c1-300
timeout 300 python scripts/r05/linear_bwd_bench.py > $O/linear_bwd_bench.txt 2>&1; cat $O/linear_bwd_bench.txt | tail -12
TZR_TUNABLE_SAVE=$O/tunableop_tuned.csv timeout 900 python scripts/r05/din_step.py 30 jagged > $O/din_step.txt 2>&1; grep din_towers $O/din_step.txt || tail -20 $O/din_step.txt
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/r05/din_step.py 20 jagged > $O/trace.log 2>&1; echo "trace rc=$?"; grep din_towers $O/trace.log
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_din_jagged.csv
rm -rf $O/trace
head -32 $O/kernel_stats_din_jagged.csv | cut -d, -f1-7 | cut -c1-160
