"""top-n kernel timing at the config-5 shape (run under rocprofv3 --kernel-trace --stats)"""
import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from tensorflowasr_amd.models import ctc_prefix_beam_decode
torch.manual_seed(0)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 9160
logits = (torch.randn(16, 750, V, device="cuda") * 3).contiguous()
counts = torch.full((16,), 750, dtype=torch.int32)
for _ in range(4):
    ctc_prefix_beam_decode(logits, counts, beam_width=1, cutoff_prob=0.99, cutoff_top_n=40, is_logits=True)
torch.cuda.synchronize()
