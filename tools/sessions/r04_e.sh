#!/bin/bash
# round 4, session e: full GPU suite on the pruned build + beam timing + config-5 bench line
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
timeout 300 python tools/r03_beamprof.py > $O/beamprof.log 2>&1; grep -a "beam\|clocks" $O/beamprof.log | cut -c1-400
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench5.json 2> $O/bench5.err; echo bench5 rc=$?
python - <<PY
import json
try:
    j=json.loads(open("$O/bench5.json").read().strip().splitlines()[-1])
    print("config5 ms/step", j["ms_per_step"], "predict", j["ms_predict"], "beam10", j["ms_beam10"], {k:(v["ms_per_step"], v["launches_per_step"]) for k,v in j["kernels"].items()})
except Exception as e: print("ERR", e, open("$O/bench5.err").read()[-600:])
PY
