#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/batch_sweep.py 1,2,3,4,8,16,17,24,32,48,64,128,256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_batch_sweep.json
cat gpurun_out/r06_batch_sweep.json
