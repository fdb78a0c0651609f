#!/bin/bash
# PMC passes of the hash variant (M1 batch): HBM bytes and L2 hit rate per kernel -> gpurun_out/pmc_hash.json (copy to profiles/)
cd $GRAFT_REPO_ROOT
NGM_MATMUL=auto NGM_CHECK=time_hash_m1 bash tools/pmc_pass.sh hf FETCH_SIZE
NGM_MATMUL=auto NGM_CHECK=time_hash_m1 bash tools/pmc_pass.sh hw WRITE_SIZE
NGM_MATMUL=auto NGM_CHECK=time_hash_m1 bash tools/pmc_pass.sh ht TCC_HIT_sum TCC_MISS_sum
python tools/pmc_hash.py $(find gpurun_out/pmc_hf -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_hw -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_ht -name "*counter_collection.csv" | head -1) > gpurun_out/pmc_hash.json
cat gpurun_out/pmc_hash.json | head -60
