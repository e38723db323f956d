#!/bin/bash
set -u
O=gpurun_out/r03h; mkdir -p $O
python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "config5_batch16" -s > $O/t1.log 2>&1
tail -8 $O/t1.log
python bench.py --no-cpu-baseline --steps 10 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03h/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"])
c=j["config5"]; print({a:b for a,b in c.items() if a not in ("kernels",)})
for n,v in sorted(c.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:6]: print("   ", n, v)
PY
