#!/bin/bash
mkdir -p gpurun_out
for on in 1 0; do
MI355ASR_GEMM256=$on timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -12
import os, sys, numpy as np, torch
sys.path.insert(0, "tests")
from helpers import co, waves
from tensorflowasr_amd.models import CTCDecoder, StreamingConformerEncoder
cfg = dict(co.STREAMING_S)
B, chunk, hist, d, V = 64, 8000, 20, 256, 1332
w = co.encoder_weights(cfg, seed=53)
w.update(co.ctc_decoder_weights(cfg, V, seed=54))
enc = StreamingConformerEncoder(dmodel=d, reduction_factor=4, num_blocks=4, head_size=64, num_heads=4, kernel_size=5, fc_factor=0.5,
                                sample_rate=16000, n_mels=80, stride_ms=10, mel_layer_type="Melspectrogram", gemm_dtype="bfloat16")
enc.add_chunk_size(chunk, 80, 640)
enc.load_weights({k: v for k, v in w.items() if not k.startswith(("project/", "decoder_conformer_block_", "fully_connected/"))}, by_name=False)
ctc = CTCDecoder(num_classes=V, dmodel=d, num_blocks=1, head_size=64, num_heads=4, kernel_size=32, fc_factor=0.5, gemm_dtype="bfloat16")
ctc.load_weights({k: v for k, v in w.items() if k.startswith(("project/", "decoder_conformer_block_", "fully_connected/"))}, by_name=False)
x = waves(B, chunk, 400)
history = np.random.default_rng(7).standard_normal((B, (hist - 1) * 13, d)).astype(np.float32)
e = enc(x)
h = torch.cat([torch.from_numpy(history).to(e.device), e], 1)
print("h contiguous", h.is_contiguous(), h.shape, "nan in h", torch.isnan(h).sum().item())
logits, amax = ctc(h, return_argmax=True)
lg, am = logits.cpu().numpy(), amax.cpu().numpy()
print("GEMM256", os.environ["MI355ASR_GEMM256"], "nan", np.isnan(lg).sum(), "amax==argmax", np.array_equal(am, lg.argmax(-1)), "==frame_argmax", np.array_equal(am, co.frame_argmax(lg)))
bad = np.argwhere(am != lg.argmax(-1))
print("mismatches", len(bad), bad[:8].tolist())
for (i, j) in bad[:6]:
    row = lg[i, j]; k1, k2 = am[i, j], row.argmax()
    print(i, j, "kernel", k1, row[k1], "numpy", k2, row[k2])
PY
done
