#!/bin/bash
# round 4, session h4: chain256 as the default of bf16 mode / dmodel 256
O=gpurun_out/r04h4; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "bf16 or config3 or streaming" > $O/tests.log 2>&1; echo tests rc=$?; tail -6 $O/tests.log; grep "chain vs layer" $O/tests.log
timeout 300 python tools/config3_only.py 40 > $O/c3.json 2> $O/c3.err; echo c3 rc=$?
python - <<PY
import json
j = json.loads(open("$O/c3.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], {n: (v["ms_per_step"], v["launches_per_step"]) for n, v in j["kernels"].items()})
PY
