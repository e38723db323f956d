#!/bin/bash
# round 4, session h2: true durations (rocprofv3 kernel trace) of chain256_bf16_kernel at 832 and 16 640 rows
O=gpurun_out/r04h2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MI355ASR_CHAIN256=2 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/tools/config3_only.py 10 > $R/$O/run.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:14]:
    print(r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
f2=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open("$f2")):
    n = r["Kernel_Name"]
    if "chain256" in n:
        g = int(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", 0))
        d[(n[-40:], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(d.items()):
    v.sort()
    print(k, len(v), "median ns", v[len(v) // 2], "min", v[0])
PY
