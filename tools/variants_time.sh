#!/bin/bash
# on the GPU box: time the torch-free harness (M1 batch, fwd / bwd ms) with every named variant library, twice, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for n in "$@"; do
    NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_$n.so NGM_MATMUL=${NGM_MATMUL:-auto} python tools/gpu_check.py ${NGM_CHECK:-time} --out=/tmp/t.json 2>&1 | grep "time_" | sed "s/^/[$n] /" | cut -c1-120
  done
done
