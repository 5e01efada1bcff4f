#!/bin/bash
O=gpurun_out/${1:-r04ao}; export O; mkdir -p $O
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary 2>> $O/bench.err | tail -1 > $O/bench_b65536.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --gemm-wgrad 2>> $O/bench.err | tail -1 > $O/bench_b65536_gemm_wgrad.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 2>> $O/bench.err | tail -1 > $O/bench_b8192.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['O']+'/bench_*.json')):
    try:
        j=json.load(open(f)); print(os.path.basename(f), round(j['value']/1e6,2), round(j['ms_per_step'],4))
    except Exception as e: print(f, 'ERR', e)
PY
