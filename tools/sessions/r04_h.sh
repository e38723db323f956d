#!/bin/bash
# round 4, session h: dmodel-256 bf16 FFModule / ConvModule tail as one launch each (chain256_bf16_kernel), config 3
O=gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "bf16 or config3" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for v in 1 0 2; do
  MI355ASR_CHAIN256=$v timeout 300 python tools/config3_only.py 40 > $O/c3_chain$v.json 2> $O/c3_chain$v.err; echo c3 chain=$v rc=$?
done
python - <<PY
import json
for t in ("chain1", "chain0", "chain2"):
    try:
        j = json.loads(open("$O/c3_%s.json" % t).read().strip().splitlines()[-1])
        k = j["kernels"]
        print(t, "ms/step", j["ms_per_step"], {n: (v["ms_per_step"], v["launches_per_step"]) for n, v in k.items() if "ffn" in n or "conv_tail" in n})
    except Exception as e:
        print(t, "ERR", e, open("$O/c3_%s.err" % t).read()[-800:])
PY
