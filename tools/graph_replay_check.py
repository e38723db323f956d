"""Does a hipGraph help?  Captures ConformerCTC.recognize() (the C-ABI call only enqueues kernels on the current stream, so
torch.cuda.graph can capture it) and times graph replay against plain stream launches for B = 1 .. 64 x 10 s; also
checks that the replayed ids equal the plain ones.  Result on an MI355X (round 2): identical ids, identical times
(B = 1: 1.408 vs 1.413 ms, B = 64: 2.843 vs 2.849 ms) -- the launch queue is never empty, the kernels themselves are
the time.  python tools/graph_replay_check.py"""
import sys, time, json, torch
sys.path.insert(0, ".")
from tensorflowasr_amd.models import ConformerCTC
L = 160000
m = ConformerCTC(1332); m._build()
res = {}
for B in (1, 2, 4, 16, 64):
    x = torch.randn(B, L, device="cuda:0") * 0.1
    m.prepare(B, L)
    for _ in range(3): ids, lens = m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize()
    ref = ids.clone(), lens.clone()
    t0 = time.perf_counter()
    for _ in range(20): m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize(); plain = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m.recognize(x, reuse_buffers=True)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            gi, gl = m.recognize(x, reuse_buffers=True)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        ok = bool((gi == ref[0]).all() and (gl == ref[1]).all())
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 20
        res[B] = {"plain_ms": round(plain * 1e3, 3), "graph_ms": round(gr * 1e3, 3), "same_ids": ok}
    except Exception as e:
        res[B] = {"plain_ms": round(plain * 1e3, 3), "error": repr(e)[:200]}
print(json.dumps(res))
