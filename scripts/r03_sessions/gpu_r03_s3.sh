#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03c}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
# round 2's tree (a7b2e11), its own library: does the side-stream plan failure reproduce on this box?
( cd _r02tree && timeout 600 python scripts/zipf_debug.py 60 0:0 > $O/zipf_debug_r02tree_v0.txt 2>&1; echo "r02 tree zipf_debug rc=$?" )
grep -v "variant 0 iter [0-9]* t=" $O/zipf_debug_r02tree_v0.txt | cut -c1-250 | tail -12
# this tree with the general forward kernel (what ran beside the plan in round 2)
TZR_TUNE=fwd_variant=1 timeout 600 python scripts/zipf_debug.py 60 0:0 > $O/zipf_debug_fwdgen_v0.txt 2>&1; echo "zipf_debug fwd_variant=1 rc=$?"
grep -v "variant 0 iter [0-9]* t=" $O/zipf_debug_fwdgen_v0.txt | cut -c1-250 | tail -12
