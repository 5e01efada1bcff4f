#!/bin/bash
# the plan's workgroups inside the forward's launch (tzr_pooled_fwd_cells_plan) against the two launches: driver flags, interleaved
O=gpurun_out/fwdplan; mkdir -p $O; rm -f $O/*.json
for rep in 1 2; do
  TZR_FWD_PLAN=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>> $O/err | tail -1 > $O/two_$rep.json
  for ord in 0 1 2; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --tune fwd_plan_order=$ord 2>> $O/err | tail -1 > $O/one_ord${ord}_$rep.json
  done
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/fwdplan/*.json')):
    try:
        d=json.load(open(p)); print(p.split('/')[-1], round(d['ms_per_step'],4))
    except Exception as e: print(p, 'ERR', e)
PY
grep -v amdgpu.ids $O/err | tail -5
