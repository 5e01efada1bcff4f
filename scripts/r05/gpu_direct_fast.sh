#!/bin/bash
# the one-launch backward with the fast tile loop (tzr_tune bwd_apply_fast: 0 = on, -1 = the general loop): batch 8192 step and
# embedding stages, alternating on one box; the 1-rank sharded proxy at 8192 per rank
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05am}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_zch_parity.py tests/test_sharded_gpu.py -m gpu -x -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1
for rep in 1 2 3; do for f in -1 0; do
TZR_TUNE=bwd_apply_fast=$f timeout 300 python bench.py --global-batch 8192 --steps 200 --no-cpu-baseline --no-e2e --no-secondary > $O/bench_b8192.fast$f.$rep.json 2>> $O/bench.err
python - $O/bench_b8192.fast$f.$rep.json $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("bwd_apply_fast", sys.argv[2], "ms_per_step", round(d["ms_per_step"],4), {k: round(v,4) if isinstance(v,float) else v for k,v in d["embedding"].items() if k in ("fwd_ms","bwd_apply_ms","frac_of_8TBps")})
PY
done; done
for f in -1 0; do
TZR_TUNE=bwd_apply_fast=$f timeout 400 python bench.py --gpus 1 --force-sharded --replicate-small --global-batch 8192 --steps 200 --warmup 12 --no-cpu-baseline --no-e2e --projection-world 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('proxy8192 fast $f ms_per_step', round(d['ms_per_step'],4), 'host busy', round(d.get('host_busy_ms_per_step',0),4))"
done
