#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "config3 or bf16 or chain256 or streaming or ring or argmax or conformer_m_and_l" > gpurun_out/gemm256_tests.log 2>&1
tail -6 gpurun_out/gemm256_tests.log
