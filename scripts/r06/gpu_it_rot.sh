#!/bin/bash
# (RECORD ONLY: -DIT_ROT was a temporary macro in interaction_top.hip; result in NOTES.md)
# wave priorities by rank inside the SIMD in the fused interaction forward (rotating per turn / static): variants against the product
for rep in 1 2; do
  echo "== product"; python scripts/bench_interaction_top.py 65536 2>&1 | grep -i "top_fwd" | head -3
  for n in 1 2 3; do
    echo "== itrot$n"; IT_LIB=libtzrec_hip_itrot$n.so python scripts/bench_interaction_top.py 65536 2>&1 | grep -i "top_fwd" | head -3
  done
done
