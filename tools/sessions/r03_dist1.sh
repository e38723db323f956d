#!/bin/bash
O=gpurun_out/r03t; mkdir -p $O
export MI355ASR_BENCH_FORCE_DIST=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-extra-configs --no-h2d > $O/dist_c2.json 2> $O/dist_c2.err
tail -c 300 $O/dist_c2.err; cut -c1-330 $O/dist_c2.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --config 5 --steps 5 --warmup 2 --no-cpu-baseline > $O/dist_c5.json 2> $O/dist_c5.err
tail -c 300 $O/dist_c5.err; cut -c1-420 $O/dist_c5.json
