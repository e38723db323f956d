#!/bin/bash
# round 6, final evidence on one box: the whole GPU suite (parity log; ceilings as set by the caller), smoke, tools/profile_round.sh r06
mkdir -p gpurun_out
export MI355ASR_PARITY_LOG=$PWD/gpurun_out/r06z_parity_full_suite.jsonl
: > $MI355ASR_PARITY_LOG
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r06z_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r06z_gpu_suite.log
unset MI355ASR_PARITY_LOG
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06z_smoke.log 2>&1
if [ "$1" = "profile" ]; then
  bash tools/profile_round.sh r06 > gpurun_out/profile_round_r06.log 2>&1
  cp gpurun_out/r06z_parity_full_suite.jsonl gpurun_out/profiles_r06/ 2>/dev/null
  tail -8 gpurun_out/profile_round_r06.log | cut -c1-300
fi
tail -4 gpurun_out/r06z_gpu_suite.log; tail -2 gpurun_out/r06z_smoke.log
