#!/bin/bash
R=$PWD
cd /tmp; export TMPDIR=/tmp
for env in "${@:-X=1}"; do
  rm -rf /tmp/pp
  env $env timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $R/tools/r03_predict_prof.py > /tmp/pp.log 2>&1 || tail -5 /tmp/pp.log
  echo "== $env"; find /tmp/pp -name "*kernel_stats.csv" | head -1 | xargs cat | python3 -c "
import csv,sys
for r in list(csv.DictReader(sys.stdin))[:14]: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), r['Percentage'])
"
done
