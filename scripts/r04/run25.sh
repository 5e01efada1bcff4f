#!/bin/bash
O=gpurun_out/${1:-r04ak}; export O; mkdir -p $O
for gb in 65536 32768 16384 8192; do
for mode in owned gemm; do
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch $gb --$mode-wgrad 2>> $O/bench.err | tail -1 > $O/bench_${mode}_wgrad_b$gb.json
done; done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['O']+'/bench_*.json')):
    try:
        j=json.load(open(f)); print(os.path.basename(f), round(j['value']/1e6,2), round(j['ms_per_step'],4))
    except Exception as e: print(f, 'ERR', e)
PY
