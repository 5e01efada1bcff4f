#!/bin/bash
# order of the two kinds of workgroup in the forward + plan launch at its final residency (seven per CU, 64-sample tiles): 2 = plan behind (default), 1 = in front, 0 = alternating
O=gpurun_out/fwdplanorder2; mkdir -p $O; rm -f $O/*
for rep in 1 2; do
  for ord in 2 1 0; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune fwd_plan_order=$ord 2>> $O/err | tail -1 > $O/ord${ord}_$rep.json
  done
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/fwdplanorder2/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']; print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), [round(k['launch_ms'],4) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
