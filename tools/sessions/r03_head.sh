#!/bin/bash
O=gpurun_out/${1:-head}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "config5 or chunk or head" > $O/t.log 2>&1
tail -3 $O/t.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from tensorflowasr_amd.config import load_yaml
from tensorflowasr_amd.models import ChunkConformer
from tensorflowasr_amd.synthetic import synth_batch
cfg = load_yaml("tensorflowasr_amd/configs/chunk_conformerS.yml")
m = ChunkConformer(cfg, phone=1332, txt=9160); m._build(seed=0)
wav = torch.from_numpy(synth_batch(0, 16, 480000)).cuda()
m._h.lib.mi355asr_profile_enable(m._h.ptr, 1)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    logits, counts = m.predict(wav)
    torch.cuda.synchronize()
    if i >= 2: print("predict ms", round((time.perf_counter() - t0) * 1e3, 3), flush=True)
from tensorflowasr_amd import _lib
names = _lib.KERNEL_NAMES
import ctypes
n = len(names)
ms = (ctypes.c_double * n)(); cnt = (ctypes.c_int64 * n)()
m._h.lib.mi355asr_profile_read(m._h.ptr, ms, cnt, n, 1)
for i in range(n):
    if cnt[i]: print(i, names[i] if names else "", cnt[i], round(ms[i] / cnt[i] * 1e3, 1), "us")
PY
