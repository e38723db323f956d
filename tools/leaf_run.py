"""Two recognize() steps of ConformerCTC(S) with the LEAF frontend at B x 10 s (a target for rocprofv3)."""
import sys

import torch

sys.path.insert(0, ".")
from tensorflowasr_amd.models import ConformerCTC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.randn(B, 160000, device="cuda:0") * 0.1
m = ConformerCTC(1332, mel_layer_type="leaf")
m._build()
m.prepare(B, 160000)
for _ in range(2):
    m.recognize(x)
torch.cuda.synchronize()
