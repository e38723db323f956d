#!/bin/bash
O=gpurun_out/${1:-topn}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "beam or config5 or chunk_asr or topn or top_n" > $O/t.log 2>&1
tail -3 $O/t.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tensorflowasr_amd.models import ctc_prefix_beam_decode
torch.manual_seed(0)
for V in (1332, 9160):
    logits = (torch.randn(16, 750, V, device="cuda") * 3).contiguous()
    counts = torch.full((16,), 750, dtype=torch.int32)
    for env in ("1", "0"):
        pass
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctc_prefix_beam_decode(logits, counts, beam_width=1, cutoff_prob=0.99, cutoff_top_n=40, is_logits=True)
        print("V", V, "beam 1 decode ms", round((time.perf_counter() - t0) * 1e3, 3), flush=True)
PY
