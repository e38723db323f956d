#!/bin/bash
mkdir -p gpurun_out
for slots in 2 4; do for nr in 1 2 3 4 6; do
  MI355ASR_RING_SLOTS=$slots MI355ASR_RING_HEAD_RANGES=$nr timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import json, os, sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
lib = _lib.lib()
r = bench.extra_config3(lib, torch.device("cuda:0"), with_cpu=False)
k = r["kernels"]
print(json.dumps({"slots": os.environ["MI355ASR_RING_SLOTS"], "ranges": os.environ["MI355ASR_RING_HEAD_RANGES"], "ms_per_step": r["ms_per_step"], "head": k["ctc.ctc_head"]["ms_per_step"],
                  "qkv": k["ctc.qkv"]["ms_per_step"], "glu": k["ctc.pw1_glu"]["ms_per_step"], "attn_out": k["ctc.attn_out"]["ms_per_step"], "project": k["ctc.ctc_project"]["ms_per_step"]}))
PY
done; done > gpurun_out/headsplit_config3b.log 2>&1
cat gpurun_out/headsplit_config3b.log
