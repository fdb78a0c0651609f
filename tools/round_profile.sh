#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes.
# Outputs land under gpurun_out/final/ (copy what should be judged into profiles/).
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_prof.log 2>&1 )
NGM_MATMUL=auto NGM_CHECK=time bash tools/pmc_pass.sh fetch FETCH_SIZE
NGM_MATMUL=auto NGM_CHECK=time bash tools/pmc_pass.sh write WRITE_SIZE
python tools/pmc_traffic.py $(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_write -name "*counter_collection.csv" | head -1) > $O/pmc_field_bwd.json
find $O -name "*kernel_stats.csv" | head -2
