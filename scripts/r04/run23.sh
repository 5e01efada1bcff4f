#!/bin/bash
O=gpurun_out/r04ag; mkdir -p $O
for dbg in 1 5 3; do
PROF_WG_DEBUG=$dbg PROF_KINDS=wgrad timeout 120 python scripts/bench_interaction_top.py --prof > $O/phase_clocks_wgrad_debug$dbg.txt 2>&1
echo "wg_debug $dbg"; grep -A5 "wgrad:" $O/phase_clocks_wgrad_debug$dbg.txt | tail -5; tail -1 $O/phase_clocks_wgrad_debug$dbg.txt
done
