#!/bin/bash
mkdir -p gpurun_out
{ for p in 0 1 0 1; do MI355ASR_ATTN_PAIR=$p timeout 300 python tools/ab_time.py default 2>&1 | grep -v amdgpu.ids | tail -1; done; } > gpurun_out/attnpair_ab.log 2>&1
cut -c1-700 gpurun_out/attnpair_ab.log
# bit identity of the ids + encoder output between the two
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, subprocess, sys, tempfile, numpy as np
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from tensorflowasr_amd.models import ConformerCTC
m = ConformerCTC(bench.NUM_CLASSES); m._build()
x = torch.from_numpy(bench.synth_batch(0, 64, 160000)).cuda()
e = m.encode(x)
np.save(sys.argv[1], e.cpu().numpy())
'''
with tempfile.TemporaryDirectory() as td:
    outs = []
    for p in ("0", "1"):
        f = os.path.join(td, p + ".npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, MI355ASR_ATTN_PAIR=p), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
    print("encoder output identical:", np.array_equal(outs[0], outs[1]), "max diff", float(np.abs(outs[0] - outs[1]).max()))
PY
