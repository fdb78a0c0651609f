cd $GRAFT_REPO_ROOT
NGM_CHECK=time bash tools/pmc_pass.sh sq1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
NGM_CHECK=time bash tools/pmc_pass.sh sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
NGM_CHECK=time bash tools/pmc_pass.sh sq3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM
python tools/pmc_summary.py $(find gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3 -name "*counter_collection.csv") > gpurun_out/sq_counters.txt 2>&1
head -50 gpurun_out/sq_counters.txt
