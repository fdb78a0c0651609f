#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "permuto or hash or cfg2 or cfg3 or reduced_precision" > $O/pytest_hash.txt 2>&1
tail -15 $O/pytest_hash.txt
timeout 300 python bench.py --variant hash --no-cpu-baseline > $O/bench_hash1.json 2> $O/bench_hash1.err; tail -c 400 $O/bench_hash1.json; tail -3 $O/bench_hash1.err
