"""Probe kept for the record ("tried and dropped"): the 64-utterance batch split over 1 / 2 / 4 HIP streams."""
import sys, time, ctypes, torch, numpy as np
sys.path.insert(0, ".")
from bench import S_CFG, NUM_CLASSES
from tensorflowasr_amd import _lib
from tensorflowasr_amd.models import ConformerCTC, _p
from tensorflowasr_amd.synthetic import synth_batch
m = ConformerCTC(NUM_CLASSES, **S_CFG); m._build(seed=0)
h = m._h
B, L = 64, 160000
wav = torch.from_numpy(synth_batch(0, B, L)).cuda()
T = h.out_frames(L)[1]
def make(nsplit):
    bs = B // nsplit
    n = ctypes.c_size_t(); _lib.check(h.lib.mi355asr_workspace_bytes(h.ptr, bs, L, ctypes.byref(n)))
    wss = [torch.empty(n.value, dtype=torch.uint8, device="cuda") for _ in range(nsplit)]
    ids = torch.empty((B, T), dtype=torch.int32, device="cuda"); lens = torch.empty((B,), dtype=torch.int32, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    def step():
        cur = torch.cuda.current_stream()
        for i, st in enumerate(streams):
            st.wait_stream(cur)
            x = wav[i*bs:(i+1)*bs]
            _lib.check(h.lib.mi355asr_recognize(h.ptr, _p(x), bs, L, None, _p(ids[i*bs:(i+1)*bs]), _p(lens[i*bs:(i+1)*bs]), _p(wss[i]), n.value, ctypes.c_void_p(st.cuda_stream)))
        for st in streams: cur.wait_stream(st)
        return ids, lens
    return step
ref = None
for nsplit in (1, 2, 4):
    step = make(nsplit)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ids, lens = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    r = ids.cpu().numpy().copy()
    if ref is None: ref = r
    print("streams", nsplit, "ms/step %.3f" % ms, "ids equal:", bool((r == ref).all()))
