# small batches on the one-tile-per-workgroup kernels (fused_ns.hip): parity of block / encoder / recognize at B = 1..8 against the
# build with MI355ASR_NS1_MAX_M=0 (the pair-pipelined kernels) and the oracle; then the kernel tables
python - <<'PY'
import subprocess, sys, os, json, tempfile
import numpy as np
code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves, golden_ctc_weights
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(3)
w = co.encoder_weights(cfg, seed=0); w.update(golden_ctc_weights())
m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
out = {}
for tag, B, L in (("b1", 1, 160000), ("b3", 3, 59000), ("b2s", 2, 45000)):
    x = waves(B, L, 40)
    enc = m.encode(x); lg = m.ctc_logits(enc)
    ids, lens = m.recognize(x)
    out[tag + "_enc"] = enc.cpu().numpy(); out[tag + "_lg"] = lg.cpu().numpy(); out[tag + "_ids"] = ids.cpu().numpy()
    ref = co.conformer_encoder(x[:1].astype(np.float64), w, cfg)
    print("RESULT %s %.3e" % (tag, np.abs(out[tag + "_enc"][:1] - ref).max()))
np.savez(sys.argv[1], **out)
'''
with tempfile.TemporaryDirectory() as td:
    res = {}
    for tag, extra in (("ns1", {}), ("pp", {"MI355ASR_NS1_MAX_M": "0"})):
        r = subprocess.run([sys.executable, "-c", code, os.path.join(td, tag + ".npz")], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=900)
        print(tag, [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")], r.stderr[-1500:] if r.returncode else "")
        res[tag] = np.load(os.path.join(td, tag + ".npz")) if r.returncode == 0 else None
    if res["ns1"] is not None and res["pp"] is not None:
        for k in res["ns1"].files:
            a, b = res["ns1"][k], res["pp"][k]
            print(k, "apart", float(np.abs(a.astype(np.float64) - b).max()))
PY
python tools/kernel_table.py 1 10 2 10 4 10 8 10 2>&1 | grep "^B=" | cut -c1-700
MI355ASR_NS1_MAX_M=0 python tools/kernel_table.py 1 10 4 10 2>&1 | grep "^B=" | cut -c1-700
