"""ms/step of ConformerCTC(S) recognize at B=64 x 10 s with and without the add_wav_info branch (random weights)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from tensorflowasr_amd.models import ConformerCTC  # noqa: E402

B, L = 64, 160000
x = torch.randn(B, L, device="cuda:0") * 0.1
out = {}
for wav in (False, True):
    m = ConformerCTC(1332, add_wav_info=wav)
    m.init_weights(seed=0) if hasattr(m, "init_weights") else m._build()
    m.prepare(B, L)
    for _ in range(3):
        m.recognize(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m.recognize(x)
    torch.cuda.synchronize()
    out["add_wav_info=%s" % wav] = round((time.perf_counter() - t0) * 100, 3)
print(json.dumps(out))
