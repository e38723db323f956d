#!/bin/bash
O=gpurun_out/${1:-c3}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "bf16 or streaming_block or config3 or conformer_m_and_l or translator or head_size_64 or native" > $O/t.log 2>&1
tail -4 $O/t.log
python tests/bench_configs.py --only 3 --steps 20 --c3-dtype both 2>/dev/null | tail -2
