"""Does a captured HIP graph of one recognize() step run faster than the same 32 launches issued on the stream?

    python tools/graph_probe.py [B] [seconds]      -> one JSON line: eager / graph-replay ms per step"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from tensorflowasr_amd.models import ConformerCTC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(float(sys.argv[2]) * 16000) if len(sys.argv) > 2 else 160000
dev = torch.device("cuda:0")
m = ConformerCTC(bench.NUM_CLASSES)
m._build()
wav = torch.from_numpy(bench.synth_batch(0, B, L)).to(dev)
m.prepare(B, L)


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


n = 300
eager = timed(lambda: m.recognize(wav, reuse_buffers=True), n)
ids0 = m._ids.clone()
out = {"B": B, "eager_ms": round(eager, 4)}
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m.recognize(wav, reuse_buffers=True)
    m._ids.fill_(-7)
    g.replay()
    torch.cuda.synchronize()
    out["graph_ids_equal"] = bool(torch.equal(m._ids, ids0))
    out["graph_ms"] = round(timed(g.replay, n), 4)
    out["eager_again_ms"] = round(timed(lambda: m.recognize(wav, reuse_buffers=True), n), 4)
except Exception as e:  # noqa: BLE001
    out["graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
print(json.dumps(out))
