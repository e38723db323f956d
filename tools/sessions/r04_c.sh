#!/bin/bash
# round 4, session c: the block as two launches (out_glu in the tail kernel's prologue)
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_launches or conformer_block or encoder_parity or chunk_conformer" > $O/tests.log 2>&1; echo tests rc=$?; tail -5 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-extra-configs --no-exact-leg > $O/bench.json 2> $O/bench.err; echo bench rc=$?
MI355ASR_PP_OGF=0 timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2d --no-extra-configs --no-exact-leg > $O/bench_ogf0.json 2> $O/bench_ogf0.err; echo bench0 rc=$?
python - <<PY
import json
for f in ("bench","bench_ogf0"):
    try:
        j=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms/step", j["ms_per_step"], "ev", j["ms_per_step_with_kernel_events"], {k:(v["avg_ms"], v["launches_per_step"]) for k,v in j["kernels"].items()})
    except Exception as e: print(f, "ERR", e, open("$O/%s.err"%f).read()[-500:])
PY
for b in 1 2 4; do echo "B sweep 10s default"; done
timeout 300 python tools/batch_sweep.py 1,2,3,4,8 10 > $O/sweep_default.json 2>&1; cat $O/sweep_default.json | tail -1
MI355ASR_SMALL_M=0 timeout 300 python tools/batch_sweep.py 1,2,3,4,8 10 > $O/sweep_fused.json 2>&1; tail -1 $O/sweep_fused.json
MI355ASR_SMALL_M=0 timeout 300 python tools/batch_sweep.py 1,2,4 2 > $O/sweep_fused_2s.json 2>&1; tail -1 $O/sweep_fused_2s.json
MI355ASR_SMALL_M=100000 timeout 300 python tools/batch_sweep.py 1,2,4 2 > $O/sweep_lat_2s.json 2>&1; tail -1 $O/sweep_lat_2s.json
MI355ASR_SMALL_M=0 timeout 300 python tools/batch_sweep.py 1,2 5 > $O/sweep_fused_5s.json 2>&1; tail -1 $O/sweep_fused_5s.json
MI355ASR_SMALL_M=100000 timeout 300 python tools/batch_sweep.py 1,2 5 > $O/sweep_lat_5s.json 2>&1; tail -1 $O/sweep_lat_5s.json
