cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcp_$i -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $R/gpurun_out/pmcp_$i.log 2>&1
done
python $R/tools/pmc_probe.py $R/gpurun_out/pmcp_1 $R/gpurun_out/pmcp_2 $R/gpurun_out/pmcp_3 $R/gpurun_out/pmcp_4 > $R/gpurun_out/pmc_probe.csv
cat $R/gpurun_out/pmc_probe.csv
