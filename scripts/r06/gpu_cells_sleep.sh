#!/bin/bash
# staggered starts of the residency slots in the cells apply (s_sleep) on top of the priorities: apply time per variant, one box
for rep in 1 2 3; do
  echo "== product"; python scripts/emb_ab.py --iters 40 "" 2>&1 | grep "^B " | sed -E 's/ +/ /g' | cut -c60-200
  for v in s1a s1b s2a s2b; do
    echo "== $v"; python scripts/emb_ab.py --lib torcheasyrec_amd/libtzrec_hip_$v.so --iters 40 "" 2>&1 | grep "^B " | sed -E 's/ +/ /g' | cut -c60-200
  done
done
