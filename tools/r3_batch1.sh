#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
L=$PWD/neural_graph_mapping_amd/lib
timeout 120 tools/micro/dot2c_split > $O/dot2c.txt 2>&1
tools/variants_time.sh base dot1 dot2 pk > $O/var1.txt 2>&1
NGM_LIB_PATH=$L/libngm_phase.so NGM_PHASE_TIMING=1 NGM_MATMUL=auto timeout 120 python tools/gpu_check.py time > $O/phase.txt 2>&1
NGM_LIB_PATH=$L/libngm_dot1.so timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_dot1.txt 2>&1
tail -3 $O/pytest_dot1.txt; cat $O/dot2c.txt | tail -12; cat $O/var1.txt
