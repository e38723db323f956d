MI355ASR_LIB=$PWD/tensorflowasr_amd/build/variants/$1.so timeout 300 python - <<'PY' 2>&1 | grep NS1STAMP | tail -6
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves, golden_ctc_weights
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(2)
w = co.encoder_weights(cfg, seed=0); w.update(golden_ctc_weights())
m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
xd = torch.from_numpy(waves(1, 160000)).cuda()
for _ in range(3): m.recognize(xd)
torch.cuda.synchronize()
PY
