#!/bin/bash
mkdir -p gpurun_out
{ for b in 64 1 8; do timeout 300 python tools/graph_probe.py $b 2>&1 | grep -v amdgpu.ids | tail -3; done; } > gpurun_out/graph_probe.log 2>&1
cat gpurun_out/graph_probe.log
