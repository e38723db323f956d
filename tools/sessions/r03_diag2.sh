#!/bin/bash
O=$PWD/gpurun_out/r03d; mkdir -p $O
V=$PWD/tensorflowasr_amd/build/variants
R=$PWD
run() {
  n=$1; shift
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2d > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/b_$n.json").read().strip().splitlines()[-1])
    k=j["kernels"]
    print("%-10s step %.3f ms  tail_ff1 %.1f us  tail_ff2 %.1f  ff1_qkv %.1f  out_glu %.1f attn %.1f subconv %.1f" % ("$n", j["ms_per_step"], k["tail_ff1"]["avg_ms"]*1e3, k["tail_ff2"]["avg_ms"]*1e3, k["ff1_qkv"]["avg_ms"]*1e3, k["out_glu"]["avg_ms"]*1e3, k["attention"]["avg_ms"]*1e3, k["subconv"]["avg_ms"]*1e3))
except Exception as e: print("$n", "ERR", e, open("$O/b_$n.err").read()[-300:])
PY
}
for dg in 0 1 2 3 8 16 24 32 40 26 27 25 10 18; do run dg$dg MI355ASR_LIB=$V/diag9.so MI355ASR_PP_DIAG=$dg; done
cd /tmp && export TMPDIR=/tmp
for dg in 0 16 27 32; do
  MI355ASR_LIB=$V/diag9.so MI355ASR_PP_DIAG=$dg timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clk_$dg -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-h2d > $O/clk_$dg.log 2>&1
  echo "== clock dg$dg"; python $R/tools/pmc_clock.py $O/clk_$dg | head -6
done
