#!/bin/bash
# round 4, session h7: chain256 final form (no split): bf16 / streaming / config-3 tests, config 3 alone, its kernel trace
O=gpurun_out/r04h7; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "bf16 or config3 or streaming or without_logits" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log; grep "chain vs layer" $O/tests.log
timeout 300 python tools/config3_only.py 40 > $O/c3.json 2> $O/c3.err; echo c3 rc=$?
python - <<PY
import json
j = json.loads(open("$O/c3.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "launches", sum(v["launches_per_step"] for v in j["kernels"].values()), j["roofline"])
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/tools/config3_only.py 10 > $R/$O/prof.log 2>&1
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/r04_config3_chain256_kernel_stats.csv; head -12 $f | cut -c1-150
