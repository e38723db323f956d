#!/bin/bash
O=gpurun_out/${1:-beam}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "beam or config5 or chunk_asr" > $O/t.log 2>&1
tail -3 $O/t.log
python tools/r03_beamprof.py 2>&1 | grep -v amdgpu.ids | tail -8
