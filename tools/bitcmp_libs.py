"""Bitwise comparison of two builds of the library: `python tools/bitcmp_libs.py LIB_A.so LIB_B.so` runs the headline model on 8 x 10 s
(encoder output, CTC logits, ids) once per library in its own process (MI355ASR_LIB; "default" = the in-tree build) and prints a
SHA-1 of every output; equal lines = bit-identical builds."""
import hashlib
import json
import os
import subprocess
import sys

if os.environ.get("BITCMP_CHILD"):
    import torch
    sys.path.insert(0, ".")
    from bench import build_model
    from tensorflowasr_amd.synthetic import synth_batch
    dev = torch.device("cuda", 0)
    m = build_model(dev, 0, 1)
    x = torch.from_numpy(synth_batch(0, 8, 160000)).to(dev)
    enc = m.encode(x)
    logits, amax = m.ctc_logits(enc, return_argmax=True)
    ids, lens = m.recognize(x)
    sha = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:16]
    print(json.dumps({"lib": os.path.basename(os.environ.get("MI355ASR_LIB", "default")), "enc": sha(enc), "logits": sha(logits), "argmax": sha(amax),
                      "ids": sha(ids)}))
else:
    for lib in sys.argv[1:] or ["default"]:
        env = dict(os.environ, BITCMP_CHILD="1")
        if lib != "default":
            env["MI355ASR_LIB"] = os.path.abspath(lib)
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-600:], flush=True)
