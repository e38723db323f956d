#!/bin/bash
set -u
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "native or conformer_block or encoder_parity or ctc_decoder_against or recognize_full or full_size or opt_in or chunk" > $O/t1.log 2>&1
tail -5 $O/t1.log
python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "config2 or config5" > $O/t2.log 2>&1
tail -3 $O/t2.log
for v in 1 0; do
MI355ASR_PP_DW=$v python bench.py --no-cpu-baseline --no-h2d > $O/bench_dw$v.json 2> $O/bench_dw$v.err
python - <<PY
import json
try:
    j=json.loads(open("$O/bench_dw$v.json").read().strip().splitlines()[-1])
    k=j.get("kernels",{})
    print("dw$v", j["ms_per_step"], {c:round(x["avg_ms"]*1e3,1) for c,x in k.items()})
except Exception as e: print("dw$v ERR", e, open("$O/bench_dw$v.err").read()[-500:])
PY
done
