#!/bin/bash
# developer tool: build a variant library that differs from lib/libngm_hip.so only in ONE source's compile flags
#   tools/variant_lib.sh NAME SOURCE.hip [extra hipcc flags...]  ->  neural_graph_mapping_amd/lib/libngm_NAME.so  (load with
#   NGM_LIB_PATH; compare on one box with tools/ab_bench.sh hip NAME).  Needs an up-to-date build (csrc/_obj/link.stamp).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
C=neural_graph_mapping_amd/csrc
stem=${src%.hip}
objs=$(cat $C/_obj/link.stamp | tr ' ' '\n' | grep -v "/${stem}-")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $C/$src -o $C/_obj/var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o neural_graph_mapping_amd/lib/libngm_$name.so $objs $C/_obj/var_$name.o
echo built libngm_$name.so
