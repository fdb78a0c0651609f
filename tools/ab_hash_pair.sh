#!/bin/bash
# A/B inside ONE library on one box: k_hash_grad launch shapes.  $1.. = "ENV=VALUE" settings to compare (each one run twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for setting in "$@"; do
  echo "== $setting rep=$rep"
  env $setting timeout 600 python bench.py --no-cpu-baseline --min-seconds 0.4 --no-aux-hash 2>gpurun_out/ab_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
a=d.get('aux_default') or {}
print('  aux_default', round(a.get('ms_per_step',0),4), a.get('kernels_us'))
" || tail -5 gpurun_out/ab_err.txt
  env $setting timeout 600 python bench.py --variant hash --no-cpu-baseline --min-seconds 0.4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('  M1-hash    ', round(d['ms_per_step'],4), d.get('kernels_us'), 'sclk', d.get('sclk_mhz'))
"
done; done
