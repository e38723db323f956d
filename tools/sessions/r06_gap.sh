#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gap
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python tools/batch_sweep.py 1 > gpurun_out/gap/sweep.log 2>&1
tail -2 gpurun_out/gap/sweep.log
python tools/gap_trace.py "gpurun_out/gap/**/*kernel_trace.csv" > gpurun_out/gap/gaps_b1.txt; cat gpurun_out/gap/gaps_b1.txt
find gpurun_out/gap -name "*kernel_trace.csv" -size +20M -delete
