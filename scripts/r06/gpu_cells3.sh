#!/bin/bash
# gpurun -- bash scripts/r06/gpu_cells3.sh <tag>: the cells plan's units in table order vs dealt round robin over the tables (same box), parity
tag=${1:-r06ac}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_pooled_parity.py tests/test_cells_plan.py tests/test_fullsize_properties.py -m gpu -x -q > $out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -2 $out/gpu_tests.log
timeout 900 python scripts/emb_ab.py --plan cells --iters 60 "cells_interleave=0" "cells_interleave=1" "cells_interleave=0" "cells_interleave=1" > $out/emb_ab_interleave.txt 2>&1; grep "^B" $out/emb_ab_interleave.txt
timeout 900 python scripts/emb_ab.py --plan cells --iters 60 --opt rowwise_adagrad "cells_interleave=0" "cells_interleave=1" >> $out/emb_ab_interleave.txt 2>&1; grep "^B" $out/emb_ab_interleave.txt | tail -2
