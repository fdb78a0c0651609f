#!/bin/bash
# same-box A/B of variant libraries on the GPU box: gpurun -- 'bash tools/ab_bench.sh hip NAME ...'  (lib/libngm_<name>.so;
# "hip" = the product library; variants from tools/variant_lib.sh).  Two runs each, interleaved.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for n in "$@"; do
  NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_$n.so timeout 300 python bench.py --no-cpu-baseline --no-aux-hash --no-aux-default > gpurun_out/ab_${n}_$rep.json 2> gpurun_out/ab_${n}_$rep.err || tail -3 gpurun_out/ab_${n}_$rep.err
done; done
python - "$@" <<'PY'
import json,sys
for n in sys.argv[1:]:
    for rep in (1,2):
        try:
            d=json.loads(open(f'gpurun_out/ab_{n}_{rep}.json').read().strip().split('\n')[-1])
            print(n, rep, round(d['ms_per_step'],4), d['kernels_us'], 'loss', d['config']['final_loss'])
        except Exception as e: print(n, rep, 'ERR', e)
PY
