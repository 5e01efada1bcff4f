#!/bin/bash
# hunt for the RCCL watchdog abort (hipErrorCapturedEvent) with the process group's flight recorder on
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03bk}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
export TORCH_SHOW_CPP_STACKTRACES=1 TORCH_NCCL_TRACE_BUFFER_SIZE=2000 TORCH_NCCL_DUMP_ON_TIMEOUT=1 TORCH_NCCL_DEBUG_INFO_TEMP_FILE=$O/nccl_trace_rank_ TORCH_NCCL_TRACE_CPP_STACK=0
fails=0
for i in $(seq 1 ${RUNS:-14}); do
  timeout 120 python -c "
import os, sys
sys.path.insert(0, '$R/tests'); sys.path.insert(0, '$R')
import test_sharded_gpu as m
m._body_whole_step_graph_world1(8192, False)
print('TZR_ISOLATED_OK', flush=True); os._exit(0)" > $O/run_$i.out 2> $O/run_$i.err
  if grep -q TZR_ISOLATED_OK $O/run_$i.out; then rm -f $O/run_$i.out $O/run_$i.err; else fails=$((fails+1)); echo "run $i FAILED"; grep -n "what()\|terminated\|Captured" $O/run_$i.err | head -5; fi
  [ $fails -ge 2 ] && break
done
echo "failures: $fails of $i runs"; ls $O | head -20
