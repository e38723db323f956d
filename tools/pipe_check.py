"""ChunkBeamPipeline against predict + beam run back to back (config 5 shape), on a fresh process."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from tensorflowasr_amd.config import load_yaml  # noqa: E402
from tensorflowasr_amd.models import ChunkBeamPipeline, ChunkConformer, ctc_prefix_beam_decode  # noqa: E402

dev = torch.device("cuda:0")
pre = sys.argv[2].split(",") if len(sys.argv) > 2 else []
if pre:                                           # stages of bench.py's main() in front, to find what disturbs the overlap
    from tensorflowasr_amd import _lib
    from tensorflowasr_amd.models import ConformerCTC
    lib = _lib.lib()
    if "sweep" in pre or "b1" in pre or "main" in pre:
        model = ConformerCTC(bench.NUM_CLASSES)
        model._build()
        w64 = torch.from_numpy(bench.synth_batch(0, 64, 160000)).to(dev)
        model.prepare(64, 160000)
        bench._timed(lambda: model.recognize(w64, reuse_buffers=True), 20)
        if "b1" in pre:
            model.prepare(1, 160000)
            bench._timed(lambda: model.recognize(w64[:1].contiguous(), reuse_buffers=True), 50)
        if "sweep" in pre:
            bench.length_sweep(model, dev)
        del model, w64
    if "cpu" in pre:
        bench.cpu_baseline_workers()
    torch.cuda.empty_cache()
    if "config3" in pre:
        bench.extra_config3(lib, dev, with_cpu="cpu3" in pre)
cfg = load_yaml(os.path.join(bench.ROOT, "tensorflowasr_amd", "configs", "chunk_conformerS.yml"))
m = ChunkConformer(cfg, phone=bench.NUM_CLASSES, txt=9160, device=dev)
m._build(seed=0)
wav = torch.from_numpy(bench.synth_batch(0, 16, 480000)).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def serial():
    lg, ct = m.predict(wav)
    return ctc_prefix_beam_decode(lg, ct, beam_width=10, cutoff_prob=0.99, cutoff_top_n=40, is_logits=True)


for _ in range(3):
    serial()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    serial()
torch.cuda.synchronize()
ts = (time.perf_counter() - t0) / n
pipe = ChunkBeamPipeline(m, beam_width=10, cutoff_prob=0.99, cutoff_top_n=40)
pipe.push(wav); pipe.push(wav)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    pipe.push(wav)
pipe.flush()
torch.cuda.synchronize()
tp = (time.perf_counter() - t0) / n
pipe.close()
print(json.dumps({"serial_ms": round(ts * 1e3, 3), "pipelined_ms": round(tp * 1e3, 3), "steps": n, "pre": pre,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("MI355ASR_")}}))
