#!/bin/bash
# Round 6: timing ablations of k_field_bwd_b3 on ONE box (what bounds the dominant kernel).  Variant libraries
# from tools/variant_lib.sh ab_<name> ngm_field_bwd_b3.hip <defines>; ablated kernels compute wrong results on purpose.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for n in hip ab_nocomp ab_nosplit ab_nosplit_noenc ab_mfmaonly ab_nomfma; do
  NGM_LIB_PATH=$PWD/neural_graph_mapping_amd/lib/libngm_$n.so timeout 300 python bench.py --min-seconds 1.5 --no-cpu-baseline --no-aux-hash --no-aux-default > gpurun_out/abl_${n}_$rep.json 2> gpurun_out/abl_${n}_$rep.err || tail -3 gpurun_out/abl_${n}_$rep.err
done; done
python - <<'PY'
import json
for n in "hip ab_nocomp ab_nosplit ab_nosplit_noenc ab_mfmaonly ab_nomfma".split():
    for rep in (1,2):
        try:
            d=json.loads(open(f'gpurun_out/abl_{n}_{rep}.json').read().strip().split('\n')[-1])
            print(n.ljust(18), rep, round(d['ms_per_step'],4), d['kernels_us'], 'sclk', d.get('sclk_mhz'))
        except Exception as e: print(n, rep, 'ERR', e)
PY
