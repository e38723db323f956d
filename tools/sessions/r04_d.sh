#!/bin/bash
# round 4, session d: two-term subsampling Dense
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "subsampling or two_launches" > $O/tests.log 2>&1; echo tests rc=$?; grep -a "subsampling Dense" $O/tests.log; tail -3 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-extra-configs --no-exact-leg > $O/bench.json 2> $O/bench.err; echo bench rc=$?
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "ev", j["ms_per_step_with_kernel_events"], "b1", j.get("latency_b1"), {k:(v["avg_ms"], v["launches_per_step"], v["scheme"]) for k,v in j["kernels"].items()})
PY
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "config2" > $O/tests2.log 2>&1; echo tests2 rc=$?; tail -3 $O/tests2.log
