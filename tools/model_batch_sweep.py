"""recognize() ms per batch of ConformerM / ConformerL, 10 s utterances, B = 1 .. 32: where the slab-ring GEMM path
(gemm_ring.hip, MI355ASR_RING_MIN_M rows) crosses the per-wave weight streams.  python tools/model_batch_sweep.py"""
import sys, time, json, torch
sys.path.insert(0, ".")
from tensorflowasr_amd.models import ConformerCTC
L = 160000
out = {}
for name, kw in (("M", dict(dmodel=256, num_blocks=13, head_size=64, num_heads=4)), ("L", dict(dmodel=512, num_blocks=13, head_size=64, num_heads=8))):
    m = ConformerCTC(1332, **kw); m._build()
    for B in (1, 2, 4, 8, 16, 32):
        x = torch.randn(B, L, device="cuda:0") * 0.1
        m.prepare(B, L)
        for _ in range(2): m.recognize(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): m.recognize(x)
        torch.cuda.synchronize()
        out["%s%d" % (name, B)] = round((time.perf_counter() - t0) / 8 * 1e3, 2)
    del m; torch.cuda.empty_cache()
print(json.dumps(out))
