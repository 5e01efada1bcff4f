#!/bin/bash
# gpurun -- bash scripts/r06/gpu_models_ab.sh <tag>: DeepFM / multi_tower_din / MMoE + ZCH steps with the MLP layers on the library's own
# tall-input kernels where they fit (default) and on the GEMM library (TZR_OWN_ROWS_GEMM=0), same box; the MLP parity test on the GPU
tag=${1:-r06af}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gemm_rows.py tests/test_config_plumbing.py tests/test_reference_module_vectors.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/gpu_tests.log
for v in 1 0 1 0; do
  TZR_OWN_ROWS_GEMM=$v timeout 900 python scripts/r05/models_step.py 30 > $out/models_own$v.txt 2>&1; echo "own=$v rc=$?"
  grep '"model"' $out/models_own$v.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  own=$v', d['model'], 'eager', round(d['ms_per_step'], 4), 'graph', d.get('graph_ms_per_step') and round(d['graph_ms_per_step'], 4), d.get('error') or '')"
done
