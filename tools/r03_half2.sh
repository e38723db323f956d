#!/bin/bash
export MI355ASR_LIB=$PWD/tensorflowasr_amd/build/variants/scdiag.so
for dg in 0 7 6 0; do
  MI355ASR_SUBCONV_DIAG=$dg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs --no-h2d 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print('diag $dg', d['ms_per_step'], {n:k[n]['avg_ms'] for n in ('subconv','tail_ff1') if n in k})
"
done
