#!/bin/bash
# (RECORD ONLY if removed) nontemporal stores of the pooled forward's output (-DFWD1_NT_STORE): default line, product vs variant, interleaved
O=gpurun_out/fwdnt; mkdir -p $O; rm -f $O/*
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>> $O/err | tail -1 > $O/product_$rep.json
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --lib libtzrec_hip_fwdnt.so 2>> $O/err | tail -1 > $O/nt_$rep.json
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob('gpurun_out/fwdnt/*.json')):
    try:
        d=json.load(open(p)); r=d['roofline']; print(p.split('/')[-1], round(d['ms_per_step'],4), round(r['frac'],4), [round(k['launch_ms'],4) for k in r['kernels']])
    except Exception as e: print(p, 'ERR', e)
PY
