#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
V=$PWD/tensorflowasr_amd/build/variants
for d in 0 1 2 4 8 16 5 7 15 31; do
  if [ $d = 0 ]; then L=""; else L="MI355ASR_LIB=$V/attn$d.so"; fi
  env $L python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d --no-extra-configs > $O/b_$d.json 2> $O/b_$d.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/b_$d.json").read().strip().splitlines()[-1]); k=j["kernels"]
    print("ADG=%-3s step %.3f attention %.1f us" % ("$d", j["ms_per_step"], k["attention"]["avg_ms"]*1e3))
except Exception as e: print("$d ERR", e, open("$O/b_$d.err").read()[-300:])
PY
done
