#!/bin/bash
# full GPU test suite + headline bench (usage: bash tools/sessions/r03_full.sh <tag>)
TAG=${1:-r03x}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
tail -4 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<PY
import json
try:
    j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    k=j.get("kernels",{})
    print(j["ms_per_step"], j["value"], {c:round(x["avg_ms"]*1e3,1) for c,x in k.items()})
    print(j.get("roofline"))
except Exception as e: print("ERR", e, open("$O/bench.err").read()[-500:])
PY
