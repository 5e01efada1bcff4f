#!/bin/bash
# the cells plan's one launch (partition, 12 us) on a side stream next to the forward lookup: fork / join inside the step's graph
O=gpurun_out/asyncplan; mkdir -p $O
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/err | tail -1 > $O/main_$rep.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --async-plan 2>> $O/err | tail -1 > $O/async_$rep.json
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/asyncplan/*.json')):
    try:
        d=json.load(open(p)); print(p.split('/')[-1], round(d['ms_per_step'],4), d.get('loss', d.get('final_loss')))
    except Exception as e: print(p, 'ERR', e)
PY
tail -5 $O/err
