#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05e}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sequence_parity.py tests/test_reference_module_vectors.py tests/test_config_plumbing.py tests/test_graph_pipeline_gpu.py -m gpu -x -q > $O/gpu_tests_din.log 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests_din.log
timeout 600 python scripts/r05/din_step.py 30 both > $O/din_step.txt 2>&1; grep din_towers $O/din_step.txt || tail -20 $O/din_step.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $O/trace -o t -- python $R/scripts/r05/din_step.py 10 jagged > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_stats.py "$DB" $O/kernel_stats_din_jagged.csv
rm -rf $O/trace
head -30 $O/kernel_stats_din_jagged.csv | cut -d, -f1-7 | cut -c1-150
