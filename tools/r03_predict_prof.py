"""ChunkConformer.predict at the config-5 shape, for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tensorflowasr_amd.config import load_yaml
from tensorflowasr_amd.models import ChunkConformer
from tensorflowasr_amd.synthetic import synth_batch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = load_yaml(os.path.join(root, "tensorflowasr_amd/configs/chunk_conformerS.yml"))
m = ChunkConformer(cfg, phone=1332, txt=9160); m._build(seed=0)
wav = torch.from_numpy(synth_batch(0, 16, 480000)).cuda()
for i in range(6):
    logits, counts = m.predict(wav)
torch.cuda.synchronize()
