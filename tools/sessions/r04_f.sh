#!/bin/bash
# round 4, session f: chunk front on the two-term subsampling conv (run-time maximum), beam install fix
O=gpurun_out/r04f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "chunk or config5 or prefix_beam" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
timeout 300 python tools/r03_beamprof.py > $O/beamprof.log 2>&1; grep -a "^beam" $O/beamprof.log
timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench5.json 2> $O/bench5.err; echo bench5 rc=$?
python - <<PY
import json
try:
    j=json.loads(open("$O/bench5.json").read().strip().splitlines()[-1])
    print("config5 ms/step", j["ms_per_step"], "predict", j["ms_predict"], "beam10", j["ms_beam10"], {k:(v["ms_per_step"], v["launches_per_step"], v["scheme"]) for k,v in j["kernels"].items()})
except Exception as e: print("ERR", e, open("$O/bench5.err").read()[-600:])
PY
