#!/bin/bash
# fixed cost of the iteration at small batches: per-phase clocks of the fused forward / backward (library built with -DNGM_PHASE_TIMING)
O=gpurun_out/r3; mkdir -p $O
NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_pt.so NGM_PHASE_TIMING=1 NGM_MATMUL=auto timeout 300 python tools/gpu_check.py time_small --out=$O/small.json > $O/small.txt 2>&1
cut -c1-1800 $O/small.txt | grep -v "^  bwd wave [1-7]\|^  wave [2-7]"
