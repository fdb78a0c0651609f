#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes,
# SQ counters, the hash variant, the evaluation path, the scene simulation and the micro-benchmarks.
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
cd $GRAFT_REPO_ROOT
NGM_MATMUL=auto NGM_CHECK=time bash tools/pmc_pass.sh sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
NGM_MATMUL=auto NGM_CHECK=time bash tools/pmc_pass.sh sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
NGM_MATMUL=auto NGM_CHECK=time bash tools/pmc_pass.sh sq3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
python tools/pmc_summary.py $(find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 -name "*counter_collection.csv") > $O/sq_counters.txt 2>&1
# the hash variant: SQ counters of its kernels + HBM bytes
NGM_MATMUL=auto NGM_CHECK=time_hash bash tools/pmc_pass.sh h1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT
NGM_MATMUL=auto NGM_CHECK=time_hash bash tools/pmc_pass.sh h2 FETCH_SIZE
NGM_MATMUL=auto NGM_CHECK=time_hash bash tools/pmc_pass.sh h3 WRITE_SIZE
python tools/pmc_summary.py $(find gpurun_out/pmc_h1 gpurun_out/pmc_h2 gpurun_out/pmc_h3 -name "*counter_collection.csv") > $O/sq_counters_hash.txt 2>&1
# HBM bytes + L2 hit rate of the hash variant's kernels (M1 batch) -> what bench.py puts into the hash rooflines' `traffic`
bash tools/profile_hash.sh > $O/pmc_hash.log 2>&1; cp gpurun_out/pmc_hash.json $O/pmc_hash.json
python tools/stage_bench.py --out $O/stage_bench.json > /dev/null 2>> $O/bench.err
python bench.py --variant hash --no-cpu-baseline > $O/bench_line_hash.json 2>> $O/bench.err
python bench.py --matmul f32 --no-cpu-baseline --no-aux-hash > $O/bench_line_f32.json 2>> $O/bench.err
python bench.py --scene-sim > $O/scene_sim.json 2>> $O/bench.err
python tools/eval_bench.py > $O/eval_bench.json 2>> $O/bench.err
# per-kernel times of the image path (render_image 640 x 480 x 640): rocprofv3 kernel stats of five renders
( cd /tmp && export TMPDIR=/tmp && NGM_EVAL_S=640 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/evalprof -o ev -- python $GRAFT_REPO_ROOT/tools/eval_bench.py > $O/evalprof.log 2>&1 )
python tools/knn_stress.py 100 > $O/knn_stress.txt 2>&1
timeout 120 tools/micro/gather_rate > $O/gather_rate.txt 2>&1
timeout 120 tools/micro/dot2c_split > $O/dot2c_split.txt 2>&1
head -30 $O/sq_counters.txt
