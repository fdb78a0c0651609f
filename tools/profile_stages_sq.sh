#!/bin/bash
# SQ counter passes over the standalone stage benchmark (tools/stage_bench.py): what the streaming stages are bound by when
# they sit below the HBM roofline -> gpurun_out/pmc_stages_sq.json (copy to profiles/)
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/stage_bench.py > /tmp/pmc_$tag.log 2>&1 ); }
rm -rf gpurun_out/pmc_q1 gpurun_out/pmc_q2 gpurun_out/pmc_q3
run q1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE
run q2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
run q3 SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS
python tools/pmc_stages_sq.py $(find gpurun_out/pmc_q1 gpurun_out/pmc_q2 gpurun_out/pmc_q3 -name "*counter_collection.csv") > gpurun_out/pmc_stages_sq.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_stages_sq.json'))['kernels']
for k,v in d.items(): print(k[:60].ljust(60), {a:(round(b,3) if isinstance(b,float) else b) for a,b in v['derived'].items()})
PY
