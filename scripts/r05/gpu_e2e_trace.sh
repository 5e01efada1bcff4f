#!/bin/bash
# the end-to-end pipeline (pinned host batches -> H2D on a copy stream -> step graph per slot), kernel by kernel incl. copies
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r05as}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o t -- python $R/bench.py --steps 4 --warmup 3 --e2e-steps 12 --no-cpu-baseline --no-secondary > $O/trace.log 2>&1; echo "trace rc=$?"
cd $R
DB=$(find $O/trace -name '*.db' | head -1)
python scripts/rocpd_timeline.py "$DB" 120 $O/timeline_e2e_tail.txt
python - "$DB" <<'PY'
import sqlite3,sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
mc=[t for t in tabs if t.startswith("rocpd_memory_copy")]
print("copy tables", mc)
if mc:
    cols=[r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
    print(cols)
    rows=cur.execute(f"select start, end, size from {mc[0]} order by start desc limit 24").fetchall()[::-1]
    for st,en,sz in rows: print(f"copy {sz:10d} B  {(en-st)/1e3:8.1f} us  start {st}")
PY
rm -rf $O/trace
tail -60 $O/timeline_e2e_tail.txt | cut -c1-130
