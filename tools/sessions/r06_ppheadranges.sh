#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_tf_goldens.py -m gpu -x -q -k "translator or chunk or config5" > gpurun_out/ppheadranges_tests.log 2>&1
tail -4 gpurun_out/ppheadranges_tests.log
for nr in 1 0 2 4 8; do
  MI355ASR_PP_HEAD_RANGES=$nr timeout 400 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import json, os, sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda:0")
r = bench.extra_config5(lib, dev, with_cpu=False)
t = bench.extra_stt(lib, dev)
print(json.dumps({"ranges": os.environ["MI355ASR_PP_HEAD_RANGES"], "c5_ms": r["ms_per_step"], "predict": r["ms_predict"], "c5_head": r["kernels"]["ctc_head"]["ms_per_step"],
                  "stt_ms": t["ms_per_step"], "tr_head": t["translator_kernels"]["ctc_head"]["ms_per_step"], "tr_kernels": t["ms_translator_kernels"]}))
PY
done > gpurun_out/ppheadranges.log 2>&1
cat gpurun_out/ppheadranges.log
