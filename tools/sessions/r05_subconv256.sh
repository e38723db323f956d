#!/bin/bash
# round 5: conv1 on the matrix pipe for dmodel 256 (the streaming configuration): parity, then the encoder step's kernels
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 -s -k "subsampling or conv_sub or streaming_block_stack or bf16_gemm_mode" > gpurun_out/sc256_tests.log 2>&1
echo "parity rc=$?" >> gpurun_out/sc256_tests.log
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_tf_goldens.py -m gpu -x -q --timeout=300 -s -k "config3 or streaming" > gpurun_out/sc256_tests2.log 2>&1
echo "config3 rc=$?" >> gpurun_out/sc256_tests2.log
O=gpurun_out/sc256_time.jsonl; : > $O
python tools/time_stream256.py >> $O 2>gpurun_out/sc256.err
MI355ASR_SUBCONV_C1M=0 python tools/time_stream256.py >> $O 2>>gpurun_out/sc256.err
timeout 300 python tests/bench_configs.py --only 3 --steps 20 --c3-dtype bf16 > gpurun_out/sc256_c3.json 2>> gpurun_out/sc256.err
grep -v "^$" gpurun_out/sc256_tests.log | tail -12; tail -4 gpurun_out/sc256_tests2.log; cat $O; tail -c 400 gpurun_out/sc256_c3.json
