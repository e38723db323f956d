#!/bin/bash
# round 4, session i: chunk path folds (front Dense, stack projections), config 5
O=gpurun_out/r04i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "layer_in_front or chunk or config5" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for v in 1 0; do
  MI355ASR_PP_PRE=$v timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench5_pre$v.json 2> $O/bench5_pre$v.err; echo bench5 pre=$v rc=$?
done
python - <<PY
import json
for t in ("pre1", "pre0"):
    try:
        j = json.loads(open("$O/bench5_%s.json" % t).read().strip().splitlines()[-1])
        print(t, "ms/step", j["ms_per_step"], "predict", j["ms_predict"], "beam10", j["ms_beam10"], {k: (v["ms_per_step"], v["launches_per_step"]) for k, v in j["kernels"].items() if k in ("ff1_qkv", "sublinear", "ctc_project", "tail_ff1")})
    except Exception as e:
        print(t, "ERR", e, open("$O/bench5_%s.err" % t).read()[-600:])
PY
