"""Per-kernel HIP-event table of the headline model at a given shape:  python tools/kernel_table.py B SECONDS [B SECONDS ...]"""
import ctypes
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tensorflowasr_amd import _lib  # noqa: E402
from tensorflowasr_amd.synthetic import synth_batch  # noqa: E402

dev = torch.device("cuda", 0)
m = bench.build_model(dev, 0, 1, False)
lib = _lib.lib()
nk = len(_lib.KERNEL_NAMES)
args = sys.argv[1:]
for B, sec in zip(args[0::2], args[1::2]):
    B, L = int(B), int(float(sec) * 16000)
    wav = torch.from_numpy(synth_batch(0, B, L)).to(dev)
    m.prepare(B, L)
    for _ in range(3):
        m.recognize(wav, reuse_buffers=True)
    _lib.check(lib.mi355asr_profile_enable(m._h.ptr, 1))
    torch.cuda.synchronize()
    ms, cnt = (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    _lib.check(lib.mi355asr_profile_read(m._h.ptr, ms, cnt, nk, 1))
    for _ in range(10):
        m.recognize(wav, reuse_buffers=True)
    torch.cuda.synchronize()
    _lib.check(lib.mi355asr_profile_read(m._h.ptr, ms, cnt, nk, 1))
    _lib.check(lib.mi355asr_profile_enable(m._h.ptr, 0))
    t = {n: [int(cnt[i]) // 10, round(1e3 * ms[i] / cnt[i], 1), round(ms[i] / 10, 3)] for i, n in enumerate(_lib.KERNEL_NAMES) if cnt[i]}
    print("B=%d %gs T=%d: total %.3f ms; (launches, us each, ms per step): %s" % (B, float(sec), L // 640, sum(v[2] for v in t.values()), json.dumps(t)))
