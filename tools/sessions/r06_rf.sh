#!/bin/bash
# reduction_factor 2 / 6 / 8: parity against the oracle + the constructor tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reduction_factors or constructor" > gpurun_out/rf_tests.log 2>&1
tail -15 gpurun_out/rf_tests.log
