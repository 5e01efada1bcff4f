#!/bin/bash
# the forward + plan launch: samples per forward workgroup (tzr_tune fwd_tile_b; 32 = default: 2048 forward workgroups at 65536)
O=gpurun_out/fwdplantile; mkdir -p $O; rm -f $O/*
for rep in 1 2; do
  for t in 0 32 56 64; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune fwd_tile_b=$t 2>> $O/err | tail -1 > $O/tile${t}_$rep.json
  done
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/fwdplantile/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']
        print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), [round(k['launch_ms'],4) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
grep -v amdgpu.ids $O/err | tail -5
