#!/bin/bash
# round 4, session h5: chain256 with four row tiles per workgroup (two hidden phases) at 16 640 rows; logits-free class head
O=gpurun_out/r04h5; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "bf16 or config3 or streaming or without_logits" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log; grep "chain vs layer" $O/tests.log
for rt in 0 2 4; do
  MI355ASR_CHAIN256_RT=$rt timeout 300 python tools/config3_only.py 40 > $O/c3_rt$rt.json 2> $O/c3_rt$rt.err; echo c3 rt=$rt rc=$?
done
python - <<PY
import json
for t in ("rt0", "rt2", "rt4"):
    try:
        j = json.loads(open("$O/c3_%s.json" % t).read().strip().splitlines()[-1])
        print(t, "ms/step", j["ms_per_step"], {n: (v["ms_per_step"], v["launches_per_step"]) for n, v in j["kernels"].items() if "ffn" in n or "conv_tail" in n or "head" in n})
    except Exception as e:
        print(t, "ERR", e, open("$O/c3_%s.err" % t).read()[-800:])
PY
