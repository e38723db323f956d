"""A / B of the N-split block kernels (fused_ns.hip) against the pair-pipelined ones (fused_pp.hip) on one box:

    python tools/ns_ab.py [B] [samples] [extra ENV=VALUE ...]

runs the S encoder + CTC decoder on the same input in two subprocesses (MI355ASR_NS=1 / 0; switches are read once per process),
prints how far the two are apart and how far each is from the fp64 oracle (first two utterances), and the per-kernel HIP-event
table of both."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, ctypes, json, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves, golden_ctc_weights
from tensorflowasr_amd import _lib
from tensorflowasr_amd.models import ConformerCTC
B, L, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cfg = small_cfg(3)
w = co.encoder_weights(cfg, seed=0); w.update(golden_ctc_weights())
m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
x = waves(B, L)
xd = torch.from_numpy(x).cuda()
enc = m.encode(xd); lg = m.ctc_logits(enc)
lib = _lib.lib(); nk = len(_lib.KERNEL_NAMES)
_lib.check(lib.mi355asr_profile_enable(m._h.ptr, 1))
for _ in range(5): m.recognize(xd)
torch.cuda.synchronize()
ms, cnt = (ctypes.c_double * nk)(), (ctypes.c_int64 * nk)()
_lib.check(lib.mi355asr_profile_read(m._h.ptr, ms, cnt, nk, 1))
for _ in range(20): m.recognize(xd)
torch.cuda.synchronize()
_lib.check(lib.mi355asr_profile_read(m._h.ptr, ms, cnt, nk, 1))
table = {n: [int(cnt[i]) // 20, round(1e3 * ms[i] / cnt[i], 2)] for i, n in enumerate(_lib.KERNEL_NAMES) if cnt[i]}
np.savez(out, enc=enc.cpu().numpy(), logits=lg.cpu().numpy())
ref_enc = co.conformer_encoder(x[:2].astype(np.float64), w, cfg); ref_lg = co.ctc_decoder(ref_enc, w, cfg)
print("RESULT " + json.dumps({"enc_err": float(np.abs(enc.cpu().numpy()[:2] - ref_enc).max()), "logits_err": float(np.abs(lg.cpu().numpy()[:2] - ref_lg).max()), "us": table}))
'''


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
    extra = dict(a.split("=", 1) for a in sys.argv[3:] if "=" in a)
    import numpy as np
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for ns in ("1", "0"):
            out = os.path.join(td, "ns%s.npz" % ns)
            r = subprocess.run([sys.executable, "-c", CODE, str(B), str(L), out], env=dict(os.environ, MI355ASR_NS=ns, **extra), capture_output=True,
                               text=True, cwd=ROOT, timeout=900)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print("NS=%s failed:\n%s" % (ns, r.stderr[-3000:]))
                return 1
            res[ns] = (json.loads(line[0][7:]), np.load(out))
        a, b = res["1"][1], res["0"][1]
        print("N-split vs pair-pipelined: encoder max|d| %.3g, logits max|d| %.3g" % (np.abs(a["enc"] - b["enc"]).max(), np.abs(a["logits"] - b["logits"]).max()))
        for ns in ("1", "0"):
            r = res[ns][0]
            print("NS=%s: vs fp64 oracle encoder %.3g logits %.3g; kernels (launches per step, us): %s" % (ns, r["enc_err"], r["logits_err"], json.dumps(r["us"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
