#!/bin/bash
# (RECORD ONLY: the knob cells_target was built for this sweep and removed; results in NOTES.md "Lock step, and what may share a launch")
# smaller units in the cells apply (more than the chip holds at once: a second round whose bounds / gather / sort run under the first
# round's tile loops): tzr_tune cells_target = expected lookups per unit (default 1076 -> 1757 units)
for rep in 1 2; do
  for t in 0 900 760 640 538 400; do
    echo "== target $t"; TZR_TUNE=cells_target=$t python scripts/emb_ab.py --iters 40 "" 2>&1 | grep "^B " | sed -E 's/ +/ /g' | cut -c60-200
  done
done
