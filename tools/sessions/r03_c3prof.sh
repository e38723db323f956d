#!/bin/bash
R=$PWD; O=$R/gpurun_out/${1:-c3p}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/tests/bench_configs.py --only 3 --steps 5 --c3-dtype bf16 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cut -c1-170 $f | head -32
