#!/bin/bash
# developer tool: build a variant library that differs from lib/libngm_hip.so only in ngm_field_bwd_b3.hip's compile flags
#   tools/b3_variant.sh NAME [extra hipcc flags...]   ->  neural_graph_mapping_amd/lib/libngm_NAME.so  (load with NGM_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=neural_graph_mapping_amd/csrc
objs=$(cat $C/_obj/link.stamp | tr ' ' '\n' | grep -v ngm_field_bwd_b3-)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c $C/ngm_field_bwd_b3.hip -o $C/_obj/b3var_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o neural_graph_mapping_amd/lib/libngm_$name.so $objs $C/_obj/b3var_$name.o
echo built libngm_$name.so
