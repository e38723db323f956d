#!/bin/bash
# round 5: first GPU session of stream256_kernel (config 3's encoder block stack in one launch)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -s -k "streaming_block_stack or bf16_gemm_mode or bf16_block" > gpurun_out/s256_tests.log 2>&1
echo "parity rc=$?" >> gpurun_out/s256_tests.log
timeout 900 python -m pytest tests/test_tf_goldens.py tests/test_gpu_baseline_shapes.py -m gpu -x -q --timeout=300 -s -k "config3 or tf or chunk" > gpurun_out/s256_tests2.log 2>&1
echo "goldens/config3 rc=$?" >> gpurun_out/s256_tests2.log
timeout 300 python tests/bench_configs.py --only 3 --steps 20 --c3-dtype bf16 > gpurun_out/s256_c3_stack.json 2> gpurun_out/s256_c3_stack.err
MI355ASR_STREAM256=0 timeout 300 python tests/bench_configs.py --only 3 --steps 20 --c3-dtype bf16 > gpurun_out/s256_c3_layers.json 2> gpurun_out/s256_c3_layers.err
tail -5 gpurun_out/s256_tests.log; tail -5 gpurun_out/s256_tests2.log
tail -c 1500 gpurun_out/s256_c3_stack.json; echo; tail -c 600 gpurun_out/s256_c3_layers.json
