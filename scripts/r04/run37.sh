#!/bin/bash
O=gpurun_out/${1:-r04ay}; export O; mkdir -p $O
timeout 300 python -m pytest tests/test_pooled_parity.py -x -q -m gpu -k "forward or fwd" > $O/gpu_tests_fwd.txt 2>&1; tail -1 $O/gpu_tests_fwd.txt
for i in 1 2; do
timeout 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-e2e --no-secondary 2>> $O/bench.err | tail -1 > $O/bench_b65536_$i.json
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ['O']+'/bench_*.json')):
    try:
        j=json.load(open(f)); r=j['roofline']; print(os.path.basename(f), round(j['value']/1e6,2), round(j['ms_per_step'],4), 'frac', round(r['frac'],3), [(k['stage'], round(k['launch_ms']*1e3,1)) for k in r['kernels']])
    except Exception as e: print(f, 'ERR', e)
PY
