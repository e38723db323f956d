#!/bin/bash
# round 5: stream256_kernel with merged LayerNorm statistics and the attention on the fp32 MFMA: parity, then timing (+ variants)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -s -k "streaming_block_stack or bf16_gemm_mode or bf16_block" > gpurun_out/s256_tests.log 2>&1
echo "parity rc=$?" >> gpurun_out/s256_tests.log
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q --timeout=300 -s -k "config3" > gpurun_out/s256_tests2.log 2>&1
echo "config3 rc=$?" >> gpurun_out/s256_tests2.log
O=gpurun_out/s256_diag2.jsonl; : > $O
python tools/time_stream256.py >> $O 2>gpurun_out/s256_diag.err
for d in 2; do MI355ASR_LIB=tools/variants/s256d$d.so python tools/time_stream256.py >> $O 2>>gpurun_out/s256_diag.err; done
tail -6 gpurun_out/s256_tests.log; tail -4 gpurun_out/s256_tests2.log
cat $O
