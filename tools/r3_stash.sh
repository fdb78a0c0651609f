#!/bin/bash
for r in 1 2 4; do
  NGM_STASH_RPW=$r timeout 300 python bench.py --no-cpu-baseline --no-aux-hash 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rpw $r', d['ms_per_step'], d['kernels_us'])"
done
