"""recognize() ms/step and audio-frames/s of ConformerCTC(S), 10 s utterances, over the batch size.

    python tools/batch_sweep.py [B1,B2,...] [seconds]     (MI355ASR_SMALL_M=<rows> moves the fused / layer-at-a-time threshold)"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from tensorflowasr_amd.models import ConformerCTC  # noqa: E402

L = int(float(sys.argv[2]) * 16000) if len(sys.argv) > 2 else 160000
m = ConformerCTC(1332)
m._build()
out = {}
BATCHES = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 8, 16, 32, 64, 128, 256]
for B in BATCHES:
    x = torch.randn(B, L, device="cuda:0") * 0.1
    m.prepare(B, L)
    for _ in range(3):
        m.recognize(x)
    torch.cuda.synchronize()
    n = 20 if B <= 64 else 8
    t0 = time.perf_counter()
    for _ in range(n):
        m.recognize(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out[B] = {"ms_step": round(dt * 1e3, 3), "audio_frames_per_s": round(B * (L // 160) / dt, 1)}
    del x
print(json.dumps(out))
