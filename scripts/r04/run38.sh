#!/bin/bash
O=gpurun_out/${1:-r04az}; export O; mkdir -p $O
timeout 600 python -m pytest tests/test_pooled_parity.py tests/test_dlrm_parity.py tests/test_sequence_parity.py tests/test_fullsize_properties.py -x -q -m gpu > $O/gpu_tests_embedding.txt 2>&1; tail -1 $O/gpu_tests_embedding.txt
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary 2>> $O/bench.err | tail -1 > $O/bench_b65536.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 2>> $O/bench.err | tail -1 > $O/bench_b8192.json
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary --global-batch 8192 --force-sharded --replicate-small 2>> $O/bench.err | tail -1 > $O/bench_sharded_proxy_b8192.json
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['O']+'/bench_*.json')):
    try:
        j=json.load(open(f)); r=j.get('roofline') or {}; print(os.path.basename(f), round(j['value']/1e6,2), round(j['ms_per_step'],4), 'frac', round(r.get('frac',0),3), [(k['stage'][:12], round(k['launch_ms']*1e3,1)) for k in r.get('kernels',[])])
    except Exception as e: print(f, 'ERR', e)
PY
