import sys, os, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import co, encoder_kwargs, maxdiff, small_cfg, waves
from tensorflowasr_amd.models import ConformerCTC
which = sys.argv[1]
base, L = (co.CONFORMER_M, 32000) if which == "M" else (co.CONFORMER_L, 24000) if which == "L" else (co.CONFORMER_S, 32000)
cfg = small_cfg(2, base)
w = co.encoder_weights(cfg, seed=41)
w.update(co.ctc_decoder_weights(cfg, 200, seed=42))
m = ConformerCTC(200, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
x = waves(3, L, 23)
enc_ref = co.conformer_encoder(x.astype(np.float64), w, cfg)
enc = m.encode(x)
print(which, os.environ.get("MI355ASR_SUBCONV_TERMS"), "enc err %.3e" % maxdiff(enc.cpu().numpy(), enc_ref), flush=True)
import torch
torch.cuda.synchronize(); print("encode ok", flush=True)
logits, amax = m.ctc_logits(enc, return_argmax=True)
torch.cuda.synchronize(); print("ctc_logits ok", flush=True)
ids, lens = m.recognize(x)
torch.cuda.synchronize(); print("recognize ok", flush=True)
