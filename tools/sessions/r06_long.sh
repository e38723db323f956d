python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=600 -k "conformer_block_parity or long_utterances or n_split" 2>&1 | tail -5
python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q --timeout=600 -k "20_s" 2>&1 | tail -3
python - <<'PY'
import torch, bench, json
dev = torch.device("cuda", 0)
m = bench.build_model(dev, 0, 1, False)
print(json.dumps(bench.length_sweep(m, dev)))
PY
MI355ASR_ATTN_LONG=0 python - <<'PY'
import torch, bench, json
dev = torch.device("cuda", 0)
m = bench.build_model(dev, 0, 1, False)
print("ATTN_LONG=0", json.dumps(bench.length_sweep(m, dev)))
PY
