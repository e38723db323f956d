#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "ring_class_head or config3 or ring_gemm or conformer_m_and_l or translator_dmodel_512" > gpurun_out/headsplit_tests.log 2>&1
tail -5 gpurun_out/headsplit_tests.log
for nr in 1 2 3 4 0; do
  MI355ASR_RING_HEAD_RANGES=$nr timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import json, os, sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
lib = _lib.lib()
r = bench.extra_config3(lib, torch.device("cuda:0"), with_cpu=False)
print(json.dumps({"ranges": os.environ["MI355ASR_RING_HEAD_RANGES"], "ms_per_step": r["ms_per_step"], "ctc_head_ms": r["kernels"]["ctc.ctc_head"]["ms_per_step"]}))
PY
done > gpurun_out/headsplit_config3.log 2>&1
cat gpurun_out/headsplit_config3.log
