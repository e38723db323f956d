"""ms/step of ConformerCTC(S) + LEAF frontend at B=64 x 10 s for the three Gabor-conv kernels (MI355ASR_LEAF_TERMS)."""
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from tensorflowasr_amd import _lib  # noqa: E402
from tensorflowasr_amd.models import ConformerCTC  # noqa: E402

B, L = 64, 160000
x = torch.randn(B, L, device="cuda:0") * 0.1
out = {}
for terms in ("0", "2", "3"):
    os.environ["MI355ASR_LEAF_TERMS"] = terms
    m = ConformerCTC(1332, mel_layer_type="leaf")
    m._build()
    m.prepare(B, L)
    for _ in range(2):
        m.recognize(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m.recognize(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 200
    h = m._h
    h.lib.mi355asr_profile_enable(h.ptr, 1)
    m.recognize(x)
    torch.cuda.synchronize()
    n = 32
    t = (ctypes.c_double * n)()
    cnt = (ctypes.c_int64 * n)()
    h.lib.mi355asr_profile_read(h.ptr, t, cnt, n, 1)
    h.lib.mi355asr_profile_enable(h.ptr, 0)
    out["terms=" + terms] = {"ms_step": round(ms, 3), "gabor_conv_ms": round(t[0], 3), "pcen_norm_ms": round(t[2], 3)}
print(json.dumps(out))
