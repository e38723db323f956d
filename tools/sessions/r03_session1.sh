#!/bin/bash
# round 3, first GPU session: parity of the pair-pipelined kernels + A/B timing against the round-2 kernels
set -u
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "native or conformer_block or encoder_parity or ctc_decoder_against or recognize_full or full_size or opt_in" > $O/t1.log 2>&1
tail -5 $O/t1.log
python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "config2" > $O/t2.log 2>&1
tail -3 $O/t2.log
python bench.py --no-cpu-baseline --no-h2d > $O/bench_pp1.json 2> $O/bench_pp1.err
MI355ASR_PP=0 python bench.py --no-cpu-baseline --no-h2d > $O/bench_pp0.json 2> $O/bench_pp0.err
python bench.py --no-cpu-baseline --no-h2d > $O/bench_pp1b.json 2> $O/bench_pp1b.err
python - <<'PY'
import json
for n in ("pp1","pp0","pp1b"):
    try:
        j=json.loads(open("gpurun_out/r03a/bench_%s.json"%n).read().strip().splitlines()[-1])
        k=j.get("kernels",{})
        print(n, j["ms_per_step"], {c:round(v["avg_ms"]*1e3,1) for c,v in k.items()})
    except Exception as e: print(n,"ERR",e)
PY
