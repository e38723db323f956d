#!/bin/bash
set -u
O=gpurun_out/r03g; mkdir -p $O
python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "config5 or ring_gemm" -s > $O/t1.log 2>&1
tail -15 $O/t1.log
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chunk" > $O/t2.log 2>&1
tail -3 $O/t2.log
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r03g/bench.json").read().strip().splitlines()[-1])
print(j["ms_per_step"], j["value"])
for k in ("config3","config5"):
    c=j.get(k,{})
    print(k, {a:b for a,b in c.items() if a not in ("kernels",)})
    for n,v in sorted(c.get("kernels",{}).items(), key=lambda kv:-kv[1]["ms_per_step"])[:8]: print("   ", n, v)
print(j.get("cpu_baseline"))
PY
python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 400 $O/bench_c5.err; head -c 600 $O/bench_c5.json
