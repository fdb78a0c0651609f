#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -x -q -m gpu > $O/pytest_fused.txt 2>&1; tail -15 $O/pytest_fused.txt
for nf in 0 1 0 1; do
  if [ $nf = 1 ]; then export NGM_NO_FUSED_COMP=1; else unset NGM_NO_FUSED_COMP; fi
  timeout 300 python bench.py --no-cpu-baseline --no-aux-hash 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no_fused=$nf', d['ms_per_step'], d['value'], d['kernels_us'], d['config'].get('final_loss'))"
done
