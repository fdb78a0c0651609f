#!/bin/bash
# round 4, call 1: correctness of the matrix-pipe transposition in k_field_bwd_b3 + same-box A/B against the round-3 library
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "stash_backward or fused_train_step_golden or full_size or fused_compositing or ragged or split or random_shapes or sparse_adam" > gpurun_out/c1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c1_tests.log
tail -5 gpurun_out/c1_tests.log
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-aux-hash > gpurun_out/c1_bench_new_$i.json 2> gpurun_out/c1_bench_new_$i.err
  NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_base_r03.so timeout 300 python bench.py --no-cpu-baseline --no-aux-hash > gpurun_out/c1_bench_base_$i.json 2> gpurun_out/c1_bench_base_$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/c1_bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, d['ms_per_step'], d['kernels_us'], d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
