#!/bin/bash
# Round 6: the complete timing ablation of k_hash_grad on ONE box.  The ablated libraries compute wrong
# results on purpose (their kernel TIMES are the point); `float` is the real opt-in mode (ngm_hash_grad_atomics) of the product
# library.  Two workloads: M1-hash (8 fields x 512 rays x 128 samples: 4 chunks per level + k_hash_reduce) and the reference's
# default iteration (32 x 512 x 24: one chunk per level, Adam of the tables inside k_hash_grad).
# build here first:
#   for v in NOSIMPLEX NOSCATTER NOMERGE; do tools/variant_lib.sh $(echo $v | tr A-Z a-z) ngm_field_bwd.hip -DNGM_ABLH_$v; done
#   tools/variant_lib.sh nosimplex_noscatter ngm_field_bwd.hip -DNGM_ABLH_NOSIMPLEX -DNGM_ABLH_NOSCATTER
#   tools/variant_lib.sh nosimplex_nomerge ngm_field_bwd.hip -DNGM_ABLH_NOSIMPLEX -DNGM_ABLH_NOMERGE
# run:  gpurun -- 'bash tools/ablate_hash_r06.sh > gpurun_out/hash_ablation.txt 2>&1'
cd $GRAFT_REPO_ROOT
line() {   # $1 = library name, $2 = atomics mode
  NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_$1.so timeout 600 python bench.py --no-cpu-baseline --min-seconds 0.4 --no-aux-hash --hash-atomics $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
a=d.get('aux_default') or {}
print('  aux_default', round(a.get('ms_per_step',0),4), a.get('kernels_us'))
"
  NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_$1.so timeout 600 python bench.py --variant hash --no-cpu-baseline --min-seconds 0.4 --hash-atomics $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('  M1-hash    ', round(d['ms_per_step'],4), d.get('kernels_us'), 'sclk', d.get('sclk_mhz'))
"
}
for rep in 1 2; do
  for cfg in "hip exact" "hip float" "nomerge exact" "nomerge float" "noscatter exact" "nosimplex exact" "nosimplex float" "nosimplex_nomerge float" "nosimplex_noscatter exact"; do
    set -- $cfg
    echo "== lib=$1 atomics=$2 rep=$rep"
    line $1 $2
  done
done
