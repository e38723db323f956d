# Stall analysis passes (SQ counters, each set in its own rocprofv3 --pmc run; never combined with trace domains):
#   bash tools/pmc_probe.sh [out_name]     -> gpurun_out/<out_name>.csv (per kernel category averages per launch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NAME=${1:-pmc_probe}
# PROBE_CMD: the command to profile (default: the headline bench)
CMD=${PROBE_CMD:-"python $R/bench.py --steps 2 --warmup 1 --min-timed-s 0 --no-cpu-baseline --no-kernel-events --no-h2d --no-extra-configs --no-exact-leg --no-latency-b1"}
i=0
DIRS=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${NAME}_p$i -- $CMD > $R/gpurun_out/${NAME}_p$i.log 2>&1
  DIRS="$DIRS $R/gpurun_out/${NAME}_p$i"
done
python $R/tools/pmc_probe.py $DIRS > $R/gpurun_out/${NAME}.csv
cat $R/gpurun_out/${NAME}.csv
