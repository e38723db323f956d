# cycle stamps of the N-split ff1_qkv kernel (workgroup 7, one line per wave and launch): tools/build_variant.py nsdN fused_ns.hip -DNS_DIAG=N (bit 4 set)
for v in "$@"; do
echo "== $v"
MI355ASR_LIB=$PWD/tensorflowasr_amd/build/variants/$v.so MI355ASR_PP_PRE=0 MI355ASR_NS=1 timeout 300 python - <<'PY' 2>&1 | grep NSSTAMP | tail -1 | python -c "
import re,sys
t=sys.stdin.read()
d={}
for k,v in re.findall(r'(\d+):(\d+)', t):
    d.setdefault(int(k), []).append(int(v))
m={k: sum(v)/len(v) for k,v in d.items() if len(v)>=3}
ks=sorted(m)
print(' '.join('%d:%d' % (k, m[k]) for k in ks))
print('deltas ' + ' '.join('%d>%d:%d' % (a,b,m[b]-m[a]) for a,b in zip(ks,ks[1:])))
"
import sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, encoder_kwargs, small_cfg, waves, golden_ctc_weights
from tensorflowasr_amd.models import ConformerCTC
cfg = small_cfg(1)
w = co.encoder_weights(cfg, seed=0); w.update(golden_ctc_weights())
m = ConformerCTC(1332, **{k: v for k, v in encoder_kwargs(cfg).items() if k != "mel_layer_type"})
m.load_weights(w, by_name=False)
xd = torch.from_numpy(waves(64, 160000)).cuda()
for _ in range(3): m.recognize(xd)
torch.cuda.synchronize()
PY
done
