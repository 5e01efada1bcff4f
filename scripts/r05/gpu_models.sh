#!/bin/bash
# the config-built model steps after the skinny-linear kernels / the windowed free-row listing: GPU tests of the touched pieces, the
# three steps (TunableOp table saved), kernel table of the MMoE + ZCH step
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05y}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dense_glue.py tests/test_config_plumbing.py tests/test_zch_parity.py tests/test_reference_module_vectors.py -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/gpu_tests.log | cut -c1-300
TZR_TUNABLE_SAVE=$O/tunableop_tuned.csv timeout 1200 python scripts/r05/models_step.py 30 > $O/models_step.txt 2>&1; grep '"model"' $O/models_step.txt | cut -c1-700 || tail -20 $O/models_step.txt
cd /tmp
TZR_TUNABLE_TUNING=0 timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/r05/models_step.py 20 mmoe_zch_b8192 > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_zch.csv
rm -rf $O/trace
head -24 $O/kernel_stats_zch.csv | cut -d, -f1-7 | cut -c1-150
