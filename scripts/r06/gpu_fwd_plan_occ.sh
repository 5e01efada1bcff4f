#!/bin/bash
# the forward + plan launch at other residencies (registers / gathers in flight): variants against the product (7 waves per SIMD, 4 gathers)
O=gpurun_out/fwdplanocc; mkdir -p $O; rm -f $O/*
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>> $O/err | tail -1 > $O/product_w7u4_$rep.json
  for v in w4u8 w5u8 w7u6 w6u4; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --lib libtzrec_hip_$v.so 2>> $O/err | tail -1 > $O/${v}_$rep.json
  done
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/fwdplanocc/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']
        print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), [round(k['launch_ms'],4) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
grep -v amdgpu.ids $O/err | tail -5
