"""which arithmetic the CTC projection ran on: logits of a fixed encoder output under the three settings (session g debugging)"""
import os, subprocess, sys, tempfile
import numpy as np
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=61)
w.update(co.ctc_decoder_weights(cfg, 300, seed=62))
m = ConformerCTC(300, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
enc = torch.from_numpy(np.random.default_rng(0).standard_normal((20, 250, 144)).astype(np.float32)).cuda()
np.save(sys.argv[1], m.ctc_logits(enc).cpu().numpy())
'''
res = {}
with tempfile.TemporaryDirectory() as td:
    for tag, extra in (("fold", {}), ("own", {"MI355ASR_PP_PRE": "0"}), ("own_f32", {"MI355ASR_PP_PRE": "0", "MI355ASR_PP_SUBLINEAR": "0"})):
        f = os.path.join(td, tag + ".npy")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **extra), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tag] = np.load(f)
for a in res:
    for b in res:
        if a < b:
            print(a, b, float(np.abs(res[a].astype(np.float64) - res[b]).max()))
