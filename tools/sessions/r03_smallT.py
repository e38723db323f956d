"""fused block path (MI355ASR_SMALL_M=0) at short utterances: block output against the fp64 oracle"""
import os, sys
os.environ["MI355ASR_SMALL_M"] = "0"
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import co, encoder_kwargs, maxdiff, small_cfg
from tensorflowasr_amd.models import ConformerEncoder
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=3)
e = ConformerEncoder(**encoder_kwargs(cfg)); e.load_weights(w, by_name=False)
for B, T in ((64, 13), (40, 30), (12, 75), (3, 100), (2, 130), (5, 64), (5, 65), (3, 17), (70, 7), (1, 250), (2, 250)):
    rng = np.random.default_rng(B * 1000 + T)
    x = rng.standard_normal((B, T, 144)).astype(np.float32)
    ref = co.conformer_block(x.astype(np.float64), w, "conformer_block_1", 36)
    got = e.conformer_block(1, x).cpu().numpy()
    print(B, T, "%.3e" % maxdiff(got, ref), flush=True)
