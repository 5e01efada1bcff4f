#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
TAG=${1:-r03d}; R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
{ echo "amdgpu module: $(cat /sys/module/amdgpu/version 2>/dev/null)"; uname -r; cat /opt/rocm/.info/version 2>/dev/null; 
  rocm-smi --showfwinfo 2>/dev/null | head -40; rocm-smi --showdriverversion 2>/dev/null | tail -3; } > $O/box_versions.txt 2>&1
head -5 $O/box_versions.txt
timeout 900 python scripts/plan_stress.py 3000 side_idle side_noise side_fwd side_apply main_noise > $O/plan_stress_3000.txt 2>&1; echo "stress rc=$?"
grep plan_stress $O/plan_stress_3000.txt
( cd _r02tree && timeout 900 python scripts/zipf_debug.py 200 0:0 > $O/zipf_debug_r02tree_v0_200.txt 2>&1; echo "r02 tree zipf_debug rc=$?" )
grep -v "variant 0 iter [0-9]* t=" $O/zipf_debug_r02tree_v0_200.txt | cut -c1-250 | tail -5
