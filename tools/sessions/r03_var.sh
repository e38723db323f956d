#!/bin/bash
# usage: bash tools/sessions/r03_var.sh <outtag> <variant names...>   (variant "base" = the regular library)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
V=$PWD/tensorflowasr_amd/build/variants
for n in "$@"; do
  if [ $n = base ]; then L=""; else L="MI355ASR_LIB=$V/$n.so"; fi
  env $L python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d --no-extra-configs > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/b_$n.json").read().strip().splitlines()[-1]); k=j["kernels"]
    print("%-10s step %.3f  %s" % ("$n", j["ms_per_step"], {c:round(x["avg_ms"]*1e3,1) for c,x in k.items() if c in ("tail_ff1","tail_ff2","ff1_qkv","out_glu","attention","subconv","stft")}))
except Exception as e: print("$n ERR", e, open("$O/b_$n.err").read()[-300:])
PY
done
