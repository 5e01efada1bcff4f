#!/bin/bash
# wave priority around the MFMA streams of the three fused interaction kernels: variants against the product library, one box
for rep in 1 2; do
  echo "== product"; python scripts/bench_interaction_top.py 65536 2>&1 | grep -i "top_fwd\|top_bwd\|wgrad" | head -8
  for n in 1 2 3; do
    echo "== itprio$n"; IT_LIB=libtzrec_hip_itprio$n.so python scripts/bench_interaction_top.py 65536 2>&1 | grep -i "top_fwd\|top_bwd\|wgrad" | head -8
  done
done
