#!/bin/bash
# round 4, session g: the layer in front of a block (subsampling Dense, CTC projection) in the block's ff1_qkv launch
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layer_in_front or opt_in_kernel_variants or two_launches or 5000_rows" > $O/tests.log 2>&1; echo tests rc=$?; tail -4 $O/tests.log
for v in 1 0; do
  MI355ASR_PP_PRE=$v timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-exact-leg --no-extra-configs > $O/bench_pre$v.json 2> $O/bench_pre$v.err; echo bench pre=$v rc=$?
done
MI355ASR_SUBLINEAR_SPLIT=2 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-exact-leg --no-extra-configs > $O/bench_b1fold.json 2> $O/bench_b1fold.err
python - <<PY
import json
for t in ("pre1", "pre0", "b1fold"):
    try:
        j = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1])
        k = j["kernels"]
        print(t, "ms/step", j["ms_per_step"], "b1", j.get("latency_b1"), {n: (v["avg_ms"], v["launches_per_step"]) for n, v in k.items() if n in ("ff1_qkv", "sublinear", "ctc_project", "tail_ff1", "subconv", "ctc_head")})
    except Exception as e:
        print(t, "ERR", e, open("$O/bench_%s.err" % t).read()[-800:])
PY
