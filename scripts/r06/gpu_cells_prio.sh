#!/bin/bash
# wave priorities by workgroup in the cells apply (the co-resident units of a CU out of lock step): stage times per mode
O=gpurun_out/cellsprio; mkdir -p $O; rm -f $O/*
for rep in 1 2; do
  for m in 0 1 6 7 8; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune cells_prio=$m 2>> $O/err | tail -1 > $O/prio${m}_$rep.json
  done
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/cellsprio/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']
        print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), [round(k['launch_ms'],4) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
grep -v amdgpu.ids $O/err | tail -5
