#!/bin/bash
mkdir -p gpurun_out
for rt in 1 2; do for cpw in 1 2 3 6; do for slots in 0 2; do
  MI355ASR_RING_RT=$rt MI355ASR_RING_CPW=$cpw MI355ASR_RING_SLOTS=$slots timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import json, os, sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
lib = _lib.lib()
r = bench.extra_config3(lib, torch.device("cuda:0"), with_cpu=False)
k = r["kernels"]
print(json.dumps({"rt": os.environ["MI355ASR_RING_RT"], "cpw": os.environ["MI355ASR_RING_CPW"], "slots": os.environ["MI355ASR_RING_SLOTS"], "ms_per_step": r["ms_per_step"], "head": k["ctc.ctc_head"]["ms_per_step"],
                  "qkv": k["ctc.qkv"]["ms_per_step"], "glu": k["ctc.pw1_glu"]["ms_per_step"], "attn_out": k["ctc.attn_out"]["ms_per_step"], "project": k["ctc.ctc_project"]["ms_per_step"]}))
PY
done; done; done > gpurun_out/ringshape_config3.log 2>&1
cat gpurun_out/ringshape_config3.log
