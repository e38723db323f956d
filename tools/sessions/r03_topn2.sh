#!/bin/bash
R=$PWD
cd /tmp; export TMPDIR=/tmp
for reg in 1 0; do
  for V in 9160 1332; do
    MI355ASR_TOPN_REG=$reg timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp_${reg}_$V -- python $R/tools/r03_topn_prof.py $V > /tmp/tp.log 2>&1 || tail -5 /tmp/tp.log
    echo "reg=$reg V=$V"; find /tmp/tp_${reg}_$V -name "*kernel_stats.csv" | head -1 | xargs grep -h "topn" | rev | cut -d, -f1-7 | rev
  done
done
