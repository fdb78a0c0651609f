#!/bin/bash
# Build variant libraries of the same sources for side-by-side timing on ONE gpu box (boxes differ by +-6 % in clock):
#   tools/variants.sh name1:"-DFLAG ..." name2:"..."     ->  neural_graph_mapping_amd/lib/libngm_<name>.so
# and time them there with   tools/variants_time.sh name1 name2 ...   (NGM_LIB_PATH selects the library; developer tools only).
cd "$(dirname "$0")/.."
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  ( NGM_HIPCC_EXTRA="$f" python -m neural_graph_mapping_amd.build --out=libngm_$n.so 2>&1 | tail -1 ) &
done
wait
