#!/bin/bash
# round 4, final verification: smoke, the full GPU suite, the batch sweep, the default bench line
O=gpurun_out/r04final; mkdir -p $O
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo smoke rc=$?; grep -a "smoke" $O/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo tests rc=$?; tail -3 $O/tests.log
timeout 600 python tools/batch_sweep.py 1,2,3,4,8,16,32,64,128,256 > $O/sweep.json 2> $O/sweep.err; tail -1 $O/sweep.json
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo bench rc=$?; grep real $O/bench.time
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"], j["roofline"]["frac"], j["latency_b1"]["ms"], j["exact_products"]["ms_per_step"], j["config3"]["ms_per_step"], j["config5"]["ms_per_step"], j["config5"]["ms_predict"])
PY
