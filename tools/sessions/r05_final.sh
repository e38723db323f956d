#!/bin/bash
# round 5, final evidence on one box: the whole GPU suite (with the excused-frame log), then tools/profile_round.sh r05h
mkdir -p gpurun_out
export MI355ASR_PARITY_LOG=$PWD/gpurun_out/r05h_parity_full_suite.jsonl
: > $MI355ASR_PARITY_LOG
timeout 1500 python -m pytest tests -m gpu -q --timeout=400 > gpurun_out/r05h_gpu_suite.log 2>&1
echo "suite rc=$?" >> gpurun_out/r05h_gpu_suite.log
unset MI355ASR_PARITY_LOG
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05h_smoke.log 2>&1
bash tools/profile_round.sh r05h > gpurun_out/profile_round_r05h.log 2>&1
cp gpurun_out/r05h_parity_full_suite.jsonl gpurun_out/profiles_r05h/
tail -4 gpurun_out/r05h_gpu_suite.log; tail -2 gpurun_out/r05h_smoke.log; tail -8 gpurun_out/profile_round_r05h.log | cut -c1-300
