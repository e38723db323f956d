#!/bin/bash
# timing-only variants of the two-term pp_block_kernel<true, true> (MI355ASR_PP_DIAG; build: tools/build_variant.py ppdiag fused_pp.hip -DMI355ASR_DIAG_KERNELS)
export MI355ASR_LIB=$PWD/tensorflowasr_amd/build/variants/ppdiag.so
export MI355ASR_PP_DW=0
for dg in 0 1 2 8 16 24 25 26 27 0; do
  MI355ASR_PP_DIAG=$dg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-h2d 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernels',{})
print('dg $dg', d['ms_per_step'], {n:k[n]['avg_ms'] for n in ('tail_ff1','tail_ff2','out_glu','attention') if n in k})
"
done
