#!/bin/bash
# where the one-rank RCCL path's extra time per step comes from: steps 10 / 50, with and without the id exchange
export MI355ASR_BENCH_FORCE_DIST=1
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps $2 --warmup 3 --no-cpu-baseline --no-extra-configs --no-h2d --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $2 $3', d['ms_per_step'])"; }
run 29521 10 gather
run 29522 50 gather
MI355ASR_BENCH_DEBUG_NO_GATHER=1 run 29523 50 nogather
unset MI355ASR_BENCH_FORCE_DIST
python bench.py --steps 50 --warmup 3 --no-cpu-baseline --no-extra-configs --no-h2d --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain steps 50', d['ms_per_step'])"
