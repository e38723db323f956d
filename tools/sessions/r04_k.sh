#!/bin/bash
# round 4, session k: band attention with the K / V window staged in LDS; config 5
O=gpurun_out/r04k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "band_attention or chunk or config5 or streaming" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for v in 1 0; do
  MI355ASR_ATTN_BAND_LDS=$v timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench5_b$v.json 2> $O/bench5_b$v.err; echo bench5 band_lds=$v rc=$?
done
python - <<PY
import json
for t in ("b1", "b0"):
    try:
        j = json.loads(open("$O/bench5_%s.json" % t).read().strip().splitlines()[-1])
        print(t, "ms/step", j["ms_per_step"], "predict", j["ms_predict"], "beam10", j["ms_beam10"], "attention", j["kernels"]["attention"])
    except Exception as e:
        print(t, "ERR", e, open("$O/bench5_%s.err" % t).read()[-600:])
PY
