#!/bin/bash
# Zipf ids: the cells plan forced (overflow units go to the worker workgroups) against the exact four-launch plan
python scripts/emb_ab.py --iters 30 --dist zipf --plan exact,cells "" 2>&1 | grep "^B " | sed -E 's/ +/ /g' | cut -c1-220
python scripts/emb_ab.py --iters 30 --dist zipf --opt rowwise_adagrad --plan exact,cells "" 2>&1 | grep "^B " | sed -E 's/ +/ /g' | cut -c1-220
