cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  (cd $R && timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcl_$i -- python tools/leaf_run.py > $R/gpurun_out/pmcl_$i.log 2>&1)
done
python $R/tools/pmc_probe.py $R/gpurun_out/pmcl_1 $R/gpurun_out/pmcl_2 $R/gpurun_out/pmcl_3 $R/gpurun_out/pmcl_4 | grep -i "kernel,\|leaf" > $R/gpurun_out/pmc_leaf.csv
cat $R/gpurun_out/pmc_leaf.csv
