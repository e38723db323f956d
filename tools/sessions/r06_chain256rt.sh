#!/bin/bash
mkdir -p gpurun_out
for rt in 0 4 5; do
  MI355ASR_CHAIN256_RT=$rt timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import json, os, sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
lib = _lib.lib()
r = bench.extra_config3(lib, torch.device("cuda:0"), with_cpu=False)
k = r["kernels"]
print(json.dumps({"rt": os.environ["MI355ASR_CHAIN256_RT"], "ms_per_step": r["ms_per_step"], "ffn": k["ctc.ffn"]["ms_per_step"], "conv_tail": k["ctc.conv_tail"]["ms_per_step"], "enc_stack": k["enc.enc_stack"]["ms_per_step"]}))
PY
done > gpurun_out/chain256rt.log 2>&1
cat gpurun_out/chain256rt.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "config3 or bf16 or chain256 or streaming" > gpurun_out/chain256rt_tests.log 2>&1
tail -4 gpurun_out/chain256rt_tests.log
