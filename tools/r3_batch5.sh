#!/bin/bash
O=gpurun_out/r3; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 1500 $O/bench_default.json; tail -3 $O/bench_default.err
timeout 300 python tools/eval_bench.py > $O/eval_bench.json 2> $O/eval_bench.err; cat $O/eval_bench.json; tail -3 $O/eval_bench.err
