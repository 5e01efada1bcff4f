#!/bin/bash
# lookups per workgroup of the one-launch backward at batch 8192 (default 256): embedding stages + step
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05ag}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for ch in 256 384 512 768 1024 256; do
TZR_TUNE=bwd_direct_ch=$ch timeout 300 python bench.py --global-batch 8192 --steps 200 --no-cpu-baseline --no-e2e --no-secondary > $O/bench_b8192.ch$ch.json 2>> $O/bench.err
python - $O/bench_b8192.ch$ch.json $ch <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("ch", sys.argv[2], "ms_per_step", round(d["ms_per_step"],4), {k: round(v,4) if isinstance(v,float) else v for k,v in d["embedding"].items()})
PY
done
