#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite at HEAD, the default line, and the 1-rank proxies of the sharded step in
# both forms (--forms ab) and with the input dist on either stream
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r06a}
O=$PWD/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench.json
for st in side main; do
  TZR_INPUT_DIST_STREAM=$st timeout 300 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 --steps 200 2>> $O/bench.err | tail -1 > $O/proxy_b8192_$st.json; echo "proxy $st rc=$?"
  python - <<PY
import json; d=json.load(open("$O/proxy_b8192_$st.json")); print("$st", d["ms_per_step"], d.get("host_queue_ms_per_step"), d.get("host_flag_wait_ms_per_step"), d["launch"][:60])
PY
done
timeout 400 python bench.py --force-sharded --replicate-small --no-cpu-baseline --global-batch 8192 --steps 100 --forms ab 2>> $O/bench.err | tail -1 > $O/proxy_b8192_forms.json; echo "forms rc=$?"
python - <<PY
import json; d=json.load(open("$O/proxy_b8192_forms.json")); print(d.get("sharded_forms")); print(d["ms_per_step"])
PY
