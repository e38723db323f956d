#!/bin/bash
# round 5: the phases of stream256_kernel without its weight stream, taken apart (timing variants; wrong results)
mkdir -p gpurun_out
O=gpurun_out/s256_diag4.jsonl; : > $O
python tools/time_stream256.py >> $O 2>gpurun_out/s256_diag.err
for f in tools/variants/s256*.so; do MI355ASR_LIB=$f python tools/time_stream256.py >> $O 2>>gpurun_out/s256_diag.err; done
cut -c1-60,150-400 $O
