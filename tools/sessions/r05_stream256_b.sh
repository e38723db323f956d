#!/bin/bash
# round 5: where stream256_kernel's time goes (timing variants; wrong results)
mkdir -p gpurun_out
O=gpurun_out/s256_diag.jsonl; : > $O
python tools/time_stream256.py >> $O 2>gpurun_out/s256_diag.err
for d in 1 2 3 6 14; do MI355ASR_LIB=tools/variants/s256d$d.so python tools/time_stream256.py >> $O 2>>gpurun_out/s256_diag.err; done
MI355ASR_STREAM256=0 python tools/time_stream256.py >> $O 2>>gpurun_out/s256_diag.err
cat $O
