#!/bin/bash
# round 4, session l: the class head behind the CTC block's tail launch
O=gpurun_out/r04l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "layer_in_front or ctc or config2 or greedy or opt_in" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for v in 1 0; do
  MI355ASR_PP_HEADF=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-exact-leg --no-extra-configs --no-latency-b1 > $O/bench_hf$v.json 2> $O/bench_hf$v.err; echo bench headf=$v rc=$?
done
python - <<PY
import json
for t in ("hf1", "hf0"):
    try:
        j = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1])
        k = j["kernels"]
        print(t, "ms/step", j["ms_per_step"], {n: (v["avg_ms"], v["launches_per_step"]) for n, v in k.items() if n in ("tail_ff2", "ctc_head", "tail_ff1", "ff1_qkv")})
    except Exception as e:
        print(t, "ERR", e, open("$O/bench_%s.err" % t).read()[-800:])
PY
