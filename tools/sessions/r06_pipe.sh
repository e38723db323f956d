#!/bin/bash
mkdir -p gpurun_out
{ for pre in main main,b1 main,sweep config3 config3,cpu3 cpu; do
  timeout 400 python tools/pipe_check.py 5 $pre 2>&1 | grep -v amdgpu.ids
done; } > gpurun_out/pipe_check.log 2>&1
cat gpurun_out/pipe_check.log | tail -20
