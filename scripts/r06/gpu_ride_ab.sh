#!/bin/bash
# (RECORD ONLY: tzr_pooled_bwd_cells_apply_adam / bench.py --no-ride were built, measured -- profiles/r06ax -- and removed in round 6)
# the dense optimizer's workgroups in the embedding apply's grid (tzr_pooled_bwd_cells_apply_adam) against the two launches: driver flags, interleaved
O=gpurun_out/ride; mkdir -p $O; rm -f $O/*
timeout 900 python -m pytest tests/test_fused_adam.py tests/test_pooled_parity.py -q -m gpu -x -k "riding or second_backward or carries or uniform1" 2>&1 | tail -2
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-ride 2>> $O/err | tail -1 > $O/two_$rep.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>> $O/err | tail -1 > $O/ride_$rep.json
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/ride/*.json')):
    try:
        d=json.load(open(p)); print(p.split('/')[-1], round(d['ms_per_step'],4), d.get('loss'), round(d['roofline']['frac'],4))
    except Exception as e: print(p, 'ERR', e)
PY
grep -v amdgpu.ids $O/err | tail -5
