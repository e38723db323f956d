#!/bin/bash
TAG=${1:-s}
R=$PWD; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-h2d --no-extra-configs > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -20
