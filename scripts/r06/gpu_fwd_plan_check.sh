#!/bin/bash
# the forward launch that carries the plan, as the default: parity tests, the default line twice against TZR_FWD_PLAN=0
O=gpurun_out/fwdplan2; mkdir -p $O; rm -f $O/*
timeout 1200 python -m pytest tests/test_pooled_parity.py tests/test_abi_errors.py tests/test_graph_pipeline_gpu.py -q -m gpu -x 2>&1 | tail -3
for rep in 1 2; do
  TZR_FWD_PLAN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>> $O/err | tail -1 > $O/two_$rep.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>> $O/err | tail -1 > $O/one_$rep.json
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/fwdplan2/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']
        print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), round(r['launch_ms'],4), [(k['kernels'][0][:36], round(k['launch_ms'],4)) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
grep -v amdgpu.ids $O/err | tail -5
