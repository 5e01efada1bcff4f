#!/bin/bash
O=gpurun_out/${1:-r04ah}; export O; mkdir -p $O
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary 2> $O/owned.err | tail -1 > $O/bench_owned_wgrad.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --gemm-wgrad 2> $O/gemm.err | tail -1 > $O/bench_gemm_wgrad.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 2>> $O/owned.err | tail -1 > $O/bench_owned_wgrad_b8192.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 --gemm-wgrad 2>> $O/gemm.err | tail -1 > $O/bench_gemm_wgrad_b8192.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('O','gpurun_out/r04ah')+'/bench_*.json')):
    try:
        j=json.load(open(f)); print(os.path.basename(f), j['value'], j['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 $O/owned.err
