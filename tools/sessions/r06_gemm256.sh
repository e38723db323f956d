#!/bin/bash
mkdir -p gpurun_out
for on in 1 0; do
  MI355ASR_GEMM256=$on timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import json, os, sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
lib = _lib.lib()
r = bench.extra_config3(lib, torch.device("cuda:0"), with_cpu=False)
k = r["kernels"]
print(json.dumps({"gemm256": os.environ["MI355ASR_GEMM256"], "ms_per_step": r["ms_per_step"], **{n: k[n]["ms_per_step"] for n in k if n.startswith("ctc.")}}))
PY
done > gpurun_out/gemm256.log 2>&1
cat gpurun_out/gemm256.log
timeout 1500 python -m pytest tests -m gpu -x -q -k "config3 or bf16 or chain256 or streaming or ring or argmax" > gpurun_out/gemm256_tests.log 2>&1
tail -6 gpurun_out/gemm256_tests.log
