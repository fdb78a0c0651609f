#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
timeout 300 python bench.py --variant hash --no-cpu-baseline > $O/bench_hash2.json 2> $O/bench_hash2.err; tail -c 300 $O/bench_hash2.json
for c in 4 8; do NGM_HASH_CHUNKS=$c timeout 300 python bench.py --variant hash --no-cpu-baseline 2>/dev/null | tail -c 230; done
timeout 300 python bench.py --no-cpu-baseline > $O/bench_f2.json 2> $O/bench_f2.err; tail -c 300 $O/bench_f2.json
