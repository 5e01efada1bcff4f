#!/bin/bash
O=gpurun_out/${1:-r04av}; export O; mkdir -p $O
timeout 600 python -m pytest tests/test_pooled_parity.py tests/test_plan_invariants.py tests/test_fullsize_properties.py -x -q -m gpu > $O/gpu_tests_embedding.txt 2>&1; tail -2 $O/gpu_tests_embedding.txt
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary 2>> $O/bench.err | tail -1 > $O/bench_b65536.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 2>> $O/bench.err | tail -1 > $O/bench_b8192.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --dist zipf 2>> $O/bench.err | tail -1 > $O/bench_zipf.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['O']+'/bench_*.json')):
    try:
        j=json.load(open(f)); r=j['roofline']; print(os.path.basename(f), round(j['value']/1e6,2), round(j['ms_per_step'],4), 'frac', round(r['frac'],3), [(k['stage'], round(k['launch_ms']*1e3,1)) for k in r['kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
