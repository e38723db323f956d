#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O
run() {  # name env...
  n=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/b_$n.json").read().strip().splitlines()[-1])
    k=j["kernels"]
    print("%-14s step %.3f ms  tail_ff1 %.1f us  tail_ff2 %.1f  ff1_qkv %.1f  out_glu %.1f attn %.1f" % ("$n", j["ms_per_step"], k["tail_ff1"]["avg_ms"]*1e3, k["tail_ff2"]["avg_ms"]*1e3, k["ff1_qkv"]["avg_ms"]*1e3, k["out_glu"]["avg_ms"]*1e3, k["attention"]["avg_ms"]*1e3))
except Exception as e: print("$n", "ERR", e, open("$O/b_$n.err").read()[-300:])
PY
}
V=$PWD/tensorflowasr_amd/build/variants
for p in 9 12 15 18; do
  run pool$p MI355ASR_LIB=$V/pool$p.so
  run pool${p}_dg24 MI355ASR_LIB=$V/pool$p.so MI355ASR_PP_DIAG=24
  run pool${p}_dg16 MI355ASR_LIB=$V/pool$p.so MI355ASR_PP_DIAG=16
done
run pool15_dg26 MI355ASR_LIB=$V/pool15.so MI355ASR_PP_DIAG=26
run pool15_dg8 MI355ASR_LIB=$V/pool15.so MI355ASR_PP_DIAG=8
run pp0 MI355ASR_LIB=$V/pool15.so MI355ASR_PP=0
