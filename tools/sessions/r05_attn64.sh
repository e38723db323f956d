#!/bin/bash
# round 5: attention_split64_kernel (head size 64, two fp16 terms): parity, then config 3
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=600 -s -k "head_size_64 or model_sizes or ring_gemm_path or bf16_gemm_mode or translator_dmodel_512 or streaming_block_stack" > gpurun_out/a64_tests.log 2>&1
echo "parity rc=$?" >> gpurun_out/a64_tests.log
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q --timeout=300 -s -k "config3" > gpurun_out/a64_tests2.log 2>&1
echo "config3 rc=$?" >> gpurun_out/a64_tests2.log
timeout 300 python tests/bench_configs.py --only 3 --steps 20 --c3-dtype bf16 > gpurun_out/a64_c3.json 2> gpurun_out/a64.err
MI355ASR_ATTN64_SPLIT=0 timeout 300 python tests/bench_configs.py --only 3 --steps 20 --c3-dtype bf16 > gpurun_out/a64_c3_old.json 2>> gpurun_out/a64.err
grep -v "^$" gpurun_out/a64_tests.log | tail -8; tail -4 gpurun_out/a64_tests2.log; tail -c 900 gpurun_out/a64_c3.json; echo; tail -c 900 gpurun_out/a64_c3_old.json
