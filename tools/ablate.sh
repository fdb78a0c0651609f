#!/bin/bash
# Timing experiments on the backward kernel: rebuild with extra hipcc defines and time the harness.
#   VARIANTS="-DA;-DB -DC;..." tools/ablate.sh      (';' separates variants; empty = the plain build)
# NGM_ABL_{NOFWD,NOWGRAD,NODGRAD,NOVALU} compile phases out of k_field_bwd16 (results are then numerically
# meaningless; only the time matters); the script forces that kernel with NGM_NO_ACT_STASH=1.
export NGM_NO_ACT_STASH=1
IFS=';' read -ra VARS <<< "${VARIANTS:- }"
for v in "${VARS[@]}"; do
  NGM_HIPCC_EXTRA="$v" python -m neural_graph_mapping_amd.build --fast > /tmp/build.log 2>&1 || { echo "build failed: $v"; tail -3 /tmp/build.log; continue; }
  echo "== [$v] $(python tools/gpu_check.py ${NGM_CHECK:-time} 2>&1 | grep time_ | cut -c1-80)"
done
