bash scripts/r06/gpu_cfg2_trace.sh 8192 | grep "adam\|finish"
bash scripts/r06/gpu_cfg2_trace.sh 65536 | grep "adam\|finish"
for rep in 1 2; do python bench.py --global-batch 8192 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b8192', d['ms_per_step'])"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b65536', d['ms_per_step'])"; done
