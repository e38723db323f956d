#!/bin/bash
# Collects the per-round rocprofv3 evidence on an MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>      e.g. r01c
# 1 kernel-trace/stats pass + 3 separate PMC passes (never combined with other trace domains), then the summary
# (profiles/<tag>_summary.md, profiles/<tag>_kernel_stats.csv, profiles/pmc_traffic.json) written into
# gpurun_out/profiles_<tag>/ for copying into profiles/.
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 5 --warmup 2 --min-timed-s 0 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-h2d --no-exact-leg --no-latency-b1"
O=$R/gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_stats -- $BENCH > $O/prof_${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${TAG}_fetch -- $BENCH > $O/prof_${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${TAG}_write -- $BENCH > $O/prof_${TAG}_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/prof_${TAG}_mfma -- $BENCH > $O/prof_${TAG}_mfma.log 2>&1
cd $R
python tools/summarize_rocprof.py $TAG $O/prof_${TAG}_stats $O/prof_${TAG}_fetch $O/prof_${TAG}_write $O/prof_${TAG}_mfma > $O/prof_${TAG}_summary.log 2>&1
mkdir -p $O/profiles_${TAG}
cp profiles/${TAG}_summary.md profiles/${TAG}_kernel_stats.csv profiles/pmc_traffic.json $O/profiles_${TAG}/
MI355ASR_PARITY_LOG=$O/profiles_${TAG}/${TAG}_parity_excused_frames.jsonl python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q -k config2 > $O/prof_${TAG}_parity.log 2>&1
python bench.py > $O/profiles_${TAG}/${TAG}_bench_n1.json 2> $O/prof_${TAG}_bench.log
# the same box's un-profiled step next to the profiler's kernel sum (round-4 review: 2.15 ms of kernels against 1.956 ms from another run)
python tools/reconcile_profile.py $O/profiles_${TAG}/${TAG}_bench_n1.json >> profiles/${TAG}_summary.md
cp profiles/${TAG}_summary.md $O/profiles_${TAG}/
head -24 profiles/${TAG}_summary.md | cut -c1-200
tail -1 $O/profiles_${TAG}/${TAG}_bench_n1.json | cut -c1-400
# BASELINE configs 3 and 5 on the same build: kernel-trace / stats of tests/bench_configs.py (their timings are in the bench line's
# config3 / config5 keys; these traces are the per-kernel evidence behind them)
cd /tmp
for c in 3 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_c$c -- python $R/tests/bench_configs.py --only $c --steps 5 --c3-dtype bf16 > $O/prof_${TAG}_c$c.log 2>&1
  f=$(find $O/prof_${TAG}_c$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/profiles_${TAG}/${TAG}_config${c}_kernel_stats.csv
  tail -2 $O/prof_${TAG}_c$c.log > $O/profiles_${TAG}/${TAG}_config${c}_bench_configs.json
done
cd $R
