#!/bin/bash
# timing-only variants of the pair-pipelined tail_ff1 kernel (library built with -DMI355ASR_DIAG_KERNELS)
O=gpurun_out/r03b; mkdir -p $O
for dg in 0 1 2 3 4 8 16 24 26 27 25 7; do
  MI355ASR_PP_DIAG=$dg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d > $O/b_$dg.json 2> $O/b_$dg.err
  python - <<PY
import json
j=json.loads(open("$O/b_$dg.json").read().strip().splitlines()[-1])
k=j["kernels"]
print("DG=%-3s step %.3f ms  tail_ff1 %.1f us  tail_ff2 %.1f  ff1_qkv %.1f" % ("$dg", j["ms_per_step"], k["tail_ff1"]["avg_ms"]*1e3, k["tail_ff2"]["avg_ms"]*1e3, k["ff1_qkv"]["avg_ms"]*1e3))
PY
done
