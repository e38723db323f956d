"""Event time of the streaming encoder's block stack (stream256_kernel, bench.py's config-3 encoder: 64 chunks x 13 rows, 4 blocks,
bf16 mode) for one build of the library (MI355ASR_LIB selects it; the timing variants of stream256.hip give wrong results):

    MI355ASR_LIB=tools/variants/s256d2.so python tools/time_stream256.py [steps]
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tensorflowasr_amd import _lib  # noqa: E402
from tensorflowasr_amd.models import StreamingConformerEncoder  # noqa: E402
from tensorflowasr_amd.synthetic import synth_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = _lib.lib()
enc = StreamingConformerEncoder(dmodel=256, reduction_factor=4, num_blocks=4, head_size=64, num_heads=4, kernel_size=5, fc_factor=0.5,
                                sample_rate=16000, n_mels=80, stride_ms=10, mel_layer_type="Melspectrogram", gemm_dtype="bfloat16")
enc.add_chunk_size(8000, 80, 640)
enc._build(seed=0)
wav = torch.from_numpy(synth_batch(0, 64, 8000)).to("cuda:0")
for _ in range(5):
    enc(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    enc(wav)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
_lib.check(lib.mi355asr_profile_enable(enc._h.ptr, 1))
for _ in range(steps):
    enc(wav)
nk = len(_lib.KERNEL_NAMES)
ms, cnt = (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
_lib.check(lib.mi355asr_profile_read(enc._h.ptr, ms, cnt, nk, 1))
print(json.dumps({"lib": os.environ.get("MI355ASR_LIB", "default"), "encoder_wall_ms": round(wall * 1e3, 4),
                  "kernels_us": {n: round(1e3 * ms[i] / steps, 2) for i, n in enumerate(_lib.KERNEL_NAMES) if cnt[i]}}))
