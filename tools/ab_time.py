"""A / B kernel times: `python tools/ab_time.py LIB_A.so LIB_B.so ...` runs the headline step (64 x 10 s) once per library
(MI355ASR_LIB selects it; "default" = the in-tree build) in its own process and prints per-category HIP-event averages and
the un-instrumented ms per step."""
import json
import os
import subprocess
import sys

if os.environ.get("AB_CHILD"):
    import ctypes
    import time
    import torch
    sys.path.insert(0, ".")
    from tensorflowasr_amd import _lib
    from bench import build_model
    from tensorflowasr_amd.synthetic import synth_batch
    dev = torch.device("cuda", 0)
    m = build_model(dev, 0, 1)
    B, L = 64, 160000
    x = torch.from_numpy(synth_batch(0, B, L)).to(dev)
    m.prepare(B, L)
    for _ in range(5):
        m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize()
    ms_step = (time.perf_counter() - t0) / 200 * 1e3
    h = m._h
    nk = len(_lib.KERNEL_NAMES)
    t, c = (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
    h.lib.mi355asr_profile_enable(h.ptr, 1)
    h.lib.mi355asr_profile_read(h.ptr, t, c, nk, 1)
    for _ in range(10):
        m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize()
    h.lib.mi355asr_profile_read(h.ptr, t, c, nk, 1)
    ids, lens = m.recognize(x)
    print(json.dumps({"lib": os.path.basename(os.environ.get("MI355ASR_LIB", "default")), "env": {k: v for k, v in os.environ.items() if k.startswith("MI355ASR_") and k != "MI355ASR_LIB"}, "ms_per_step": round(ms_step, 4),
                      "ids_checksum": int(ids.long().sum().item()), "lens_sum": int(lens.sum().item()),
                      "kernels_us": {n: round(t[i] / c[i] * 1e3, 1) for i, n in enumerate(_lib.KERNEL_NAMES) if c[i]}}))
else:
    for spec in sys.argv[1:] or ["default"]:
        lib, _, envs = spec.partition("@")                 # LIB[@NAME=VALUE,NAME=VALUE]: extra environment for that run
        env = dict(os.environ, AB_CHILD="1")
        for kv in filter(None, envs.split(",")):
            env[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
        if lib != "default":
            env["MI355ASR_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
