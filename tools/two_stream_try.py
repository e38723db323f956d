"""Does splitting the batch over two streams help?  Two ConformerCTC(S) instances, 32 x 10 s each, enqueued on two
streams (their kernels can overlap: each fills half the CUs) against one instance with 64 x 10 s.
python tools/two_stream_try.py"""
import sys, time, json, torch
sys.path.insert(0, ".")
from tensorflowasr_amd.models import ConformerCTC
L = 160000
res = {}
m64 = ConformerCTC(1332); m64._build()
x64 = torch.randn(64, L, device="cuda:0") * 0.1
m64.prepare(64, L)
for _ in range(3): m64.recognize(x64, reuse_buffers=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): m64.recognize(x64, reuse_buffers=True)
torch.cuda.synchronize(); res["one_stream_64"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
for parts in (2, 4):
    ms = [ConformerCTC(1332) for _ in range(parts)]
    for m in ms: m._build()
    xs = [x64[i * (64 // parts):(i + 1) * (64 // parts)].contiguous() for i in range(parts)]
    st = [torch.cuda.Stream() for _ in range(parts)]
    for m, x, s in zip(ms, xs, st):
        with torch.cuda.stream(s):
            m.prepare(64 // parts, L)
            for _ in range(3): m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        for m, x, s in zip(ms, xs, st):
            with torch.cuda.stream(s):
                m.recognize(x, reuse_buffers=True)
    torch.cuda.synchronize(); res["%d_streams_x%d" % (parts, 64 // parts)] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    del ms
print(json.dumps(res))
