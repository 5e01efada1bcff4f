#!/bin/bash
# config 2 (B = 8192): the one-launch direct backward against the cells plan (partition + apply) at the same batch
O=gpurun_out/cfg2cells; mkdir -p $O
for rep in 1 2; do
  timeout 300 python bench.py --global-batch 8192 --no-cpu-baseline 2>> $O/err | tail -1 > $O/direct_$rep.json
  timeout 300 python bench.py --global-batch 8192 --no-cpu-baseline --tune bwd_direct=-1 2>> $O/err | tail -1 > $O/cells_$rep.json
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/cfg2cells/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']
        print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), [(k['kernels'][0][:28], round(k['launch_ms'],4)) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
tail -5 $O/err
