#!/bin/bash
# gpurun -- bash scripts/r06/gpu_models_fuse.sh <tag>: DeepFM / multi_tower_din / MMoE + ZCH steps with the Linear + ReLU layers' bias gradients
# taken by the dense optimizer's launch as partial rows (default) and with a finishing launch per layer (TZR_MODELS_FUSE_FINISH=0), same box
tag=${1:-r06aq}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_fused_adam.py tests/test_config_plumbing.py tests/test_dense_glue.py tests/test_graph_pipeline_gpu.py -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/gpu_tests.log
for v in 1 0 1 0; do export TZR_MLP_OWN_WGRAD=${OWN_WGRAD:-1};
  TZR_MODELS_FUSE_FINISH=$v timeout 900 python scripts/r05/models_step.py 30 > $out/models_fuse$v.txt 2>&1; echo "fuse=$v rc=$?"
  grep '"model"' $out/models_fuse$v.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  fuse=$v', d['model'], 'eager', round(d['ms_per_step'], 4), 'graph', d.get('graph_ms_per_step') and round(d['graph_ms_per_step'], 4), d.get('error') or '', d.get('graph_error') or '')"
done
