#!/bin/bash
# quick check: a pytest -k selection + the headline bench's kernel table (usage: bash tools/sessions/r03_quick.sh <tag> "<pytest -k expr>")
TAG=${1:-q}; K=${2:-"native"}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "$K" > $O/t.log 2>&1
tail -4 $O/t.log
python bench.py --no-cpu-baseline --no-h2d --no-extra-configs > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    k=j.get("kernels",{})
    print(j["ms_per_step"], {c:round(x["avg_ms"]*1e3,1) for c,x in k.items()})
except Exception as e: print("ERR", e, open("$O/bench.err").read()[-500:])
PY
