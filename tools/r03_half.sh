#!/bin/bash
export MI355ASR_LIB=$PWD/tensorflowasr_amd/build/variants/ppdiag.so
export MI355ASR_PP_DW=0
for dg in 0 64 65 0; do
  MI355ASR_PP_DIAG=$dg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs --no-h2d 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print('dg $dg', d['ms_per_step'], {n:k[n]['avg_ms'] for n in ('tail_ff1','tail_ff2','ff1_qkv','dwconv') if n in k})
"
done
