#!/bin/bash
# PMC passes over the standalone stage benchmark: real HBM bytes and L2 hit rate per launch of the stage kernels
# -> gpurun_out/pmc_stages.json (copy to profiles/)
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/stage_bench.py > /tmp/pmc_$tag.log 2>&1 ); }
rm -rf gpurun_out/pmc_sf gpurun_out/pmc_sw gpurun_out/pmc_st
run sf FETCH_SIZE
run sw WRITE_SIZE
run st TCC_HIT_sum TCC_MISS_sum
NGM_PMC_KERNELS="k_encode_points,k_composite_fwd,k_composite_bwd,k_sample_rays,k_field_points_fwd,k_field_bwd16" \
NGM_PMC_COMMAND="rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} (three passes) --output-format csv -- python tools/stage_bench.py   [262144 rays x 128 samples; hash stages: 8 fields x 524288 points]" \
python tools/pmc_hash.py $(find gpurun_out/pmc_sf -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_sw -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_st -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_stages.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_stages.json'))['kernels']
for k,v in d.items(): print(k[:70].ljust(70), 'HBM MB', None if v['hbm_bytes'] is None else round(v['hbm_bytes']/1e6,1), 'L2 hit', None if v['l2_hit_rate'] is None else round(v['l2_hit_rate'],3))
PY
