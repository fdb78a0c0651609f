#!/bin/bash
# Would stashing the forward's lattice search for k_hash_grad pay?  Two timing ablations, A/B on ONE box (results of the
# ablated libraries are meaningless, their kernel times are not):
#   hcells    = the hash forward additionally writes 24 B per (sample, level) (-DNGM_ABLF_HASHCELLS: 4 x 16-bit slots + 4 fp32
#               weights, level-major, non-temporal) -- what the stash would cost the forward
#   nosimplex = k_hash_grad without permuto_simplex (-DNGM_ABLH_NOSIMPLEX), same loads as today -- an UPPER bound of what the
#               stash would save (the real kernel would read 24 instead of 8 bytes per sample-level from HBM)
# build here:   tools/variant_lib.sh hcells ngm_field_fwd.hip -DNGM_ABLF_HASHCELLS
#               tools/variant_lib.sh nosimplex ngm_field_bwd.hip -DNGM_ABLH_NOSIMPLEX
# run:          gpurun -- 'bash tools/ablate_hash_stash.sh'
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for n in hip hcells nosimplex; do
  echo "== $n $rep"
  NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_$n.so timeout 300 python bench.py --variant hash --no-cpu-baseline --min-seconds 0.5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('M1-hash', round(d['ms_per_step'],4), d.get('kernels_us'))
"
done; done
