#!/bin/bash
# same-box A/B of the embedding calls: wave priorities by residency slot.  clean = none anywhere; product = the cells apply;
# prioall = + partition, one-launch backward (B = 8192), exact apply (Zipf ids / plan=exact)
for rep in 1 2; do
  for v in clean prioall; do
    echo "== $v"; python scripts/emb_ab.py --lib torcheasyrec_amd/libtzrec_hip_$v.so --iters 40 --B 65536,8192 --plan auto,exact "" 2>&1 | grep "^B "
    python scripts/emb_ab.py --lib torcheasyrec_amd/libtzrec_hip_$v.so --iters 40 --dist zipf "" 2>&1 | grep "^B "
    python scripts/emb_ab.py --lib torcheasyrec_amd/libtzrec_hip_$v.so --iters 40 --opt rowwise_adagrad "" 2>&1 | grep "^B "
  done
  echo "== product"; python scripts/emb_ab.py --iters 40 --B 65536,8192 --plan auto,exact "" 2>&1 | grep "^B "
  python scripts/emb_ab.py --iters 40 --dist zipf "" 2>&1 | grep "^B "
  python scripts/emb_ab.py --iters 40 --opt rowwise_adagrad "" 2>&1 | grep "^B "
done
