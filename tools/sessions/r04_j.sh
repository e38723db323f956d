#!/bin/bash
# round 4, session j: one row tile per wave in the two-term subsampling conv for small grids (streaming chunks, single utterances)
O=gpurun_out/r04j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "subsampling or subconv or config3 or streaming or config2" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for rt in 0 2; do
  MI355ASR_SUBCONV_RT=$rt timeout 300 python tools/config3_only.py 40 > $O/c3_rt$rt.json 2> $O/c3_rt$rt.err
  MI355ASR_SUBCONV_RT=$rt timeout 300 python tools/batch_sweep.py 1,2,4,8 > $O/sweep_rt$rt.json 2> $O/sweep_rt$rt.err
done
python - <<PY
import json
for t in ("rt0", "rt2"):
    j = json.loads(open("$O/c3_%s.json" % t).read().strip().splitlines()[-1])
    print(t, "config3 ms/step", j["ms_per_step"], j["kernels"]["enc.subconv"])
    print(t, "sweep", open("$O/sweep_%s.json" % t).read().strip().splitlines()[-1])
PY
