"""Kernel times (HIP events via mi355asr_profile_*) under environment-selected kernel variants, one process per mode.
The `diagN` modes need a library built with MI355ASR_EXTRA_HIPCC_FLAGS=-DMI355ASR_DIAG_KERNELS (`python -m
tensorflowasr_amd.build --force`); they time wrong-result variants of the split subsampling kernel."""
import json
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import ctypes
    import torch
    sys.path.insert(0, ".")
    from tensorflowasr_amd.models import ConformerCTC
    B, L = 64, 160000
    x = torch.randn(B, L, device="cuda:0") * 0.1
    m = ConformerCTC(1332)
    m._build()
    m.prepare(B, L)
    for _ in range(3):
        m.recognize(x)
    h = m._h
    h.lib.mi355asr_profile_enable(h.ptr, 1)
    for _ in range(5):
        m.recognize(x)
    torch.cuda.synchronize()
    t = (ctypes.c_double * 32)()
    c = (ctypes.c_int64 * 32)()
    h.lib.mi355asr_profile_read(h.ptr, t, c, 32, 1)
    print(json.dumps({"mode": sys.argv[1], "subconv_ms": round(t[3] / max(c[3], 1), 4), "sublinear_ms": round(t[4] / max(c[4], 1), 4), "out_glu_ms": round(t[16] / max(c[16], 1), 4), "ff1_qkv_ms": round(t[15] / max(c[15], 1), 4),
                      "tail_ff2_ms": round(t[17] / max(c[17], 1), 4)}))
else:
    for mode in ("0", "tailring"):
        env = dict(os.environ)
        if mode == "f32":
            env["MI355ASR_SUBCONV_F32"] = "1"
        elif mode.startswith("og"):
            env["MI355ASR_OUTGLU_SPLIT"] = mode[2:]
        elif mode == "ffring":
            env["MI355ASR_FF1QKV_RING"] = "1"
        elif mode == "tailring":
            env["MI355ASR_TAILFF2_RING"] = "1"
        elif mode.startswith("diag"):
            env["MI355ASR_SUBCONV_DIAG"] = mode[4:]
        else:
            env["MI355ASR_SUBCONV_LATE"] = mode
        out = subprocess.run([sys.executable, __file__, mode], env=env, capture_output=True, text=True)
        print(out.stdout.strip().split("\n")[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
