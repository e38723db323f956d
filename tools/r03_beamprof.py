"""per-frame clock counters of the device beam search at the config-5 shape (MI355ASR_BEAM_PROF=1)"""
import os, sys, time
os.environ["MI355ASR_BEAM_PROF"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch
from tensorflowasr_amd.config import load_yaml
from tensorflowasr_amd.models import ChunkConformer, ctc_prefix_beam_decode
from tensorflowasr_amd.synthetic import synth_batch
cfg = load_yaml("tensorflowasr_amd/configs/chunk_conformerS.yml")
m = ChunkConformer(cfg, phone=1332, txt=9160); m._build(seed=0)
wav = torch.from_numpy(synth_batch(0, 16, 480000)).cuda()
logits, counts = m.predict(wav)
for beam in (10, 100):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctc_prefix_beam_decode(logits, counts, beam_width=beam, cutoff_prob=0.99, cutoff_top_n=40, is_logits=True)
        print("beam", beam, "ms", round((time.perf_counter() - t0) * 1e3, 2), flush=True)
