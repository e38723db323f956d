#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stt_trace
cat > /tmp/stt_run.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
print(bench.extra_stt(_lib.lib(), torch.device("cuda:0"))["ms_per_step"])
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stt_trace -- python /tmp/stt_run.py > gpurun_out/stt_trace/run.log 2>&1
tail -1 gpurun_out/stt_trace/run.log
python - <<'PY'
import csv, glob
rows = []
for p in glob.glob("gpurun_out/stt_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# one step = from an fft_stft kernel to the next one; take the last complete step
st = [i for i, r in enumerate(rows) if "fft_stft" in r[2]]
a, b = st[-2], st[-1]
seg = rows[a:b]
span = rows[b][0] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print("stt step: %d kernels, span %.1f us, busy %.1f us, idle %.1f us" % (len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3))
prev = seg[0]
for r in seg[1:] + [rows[b]]:
    g = (r[0] - prev[1]) / 1e3
    if g > 3:
        print("  gap %.1f us before %s (after %s)" % (g, r[2].replace("(anonymous namespace)::", "")[:70], prev[2].replace("(anonymous namespace)::", "")[:50]))
    prev = r
PY
find gpurun_out/stt_trace -name "*kernel_trace.csv" -size +30M -delete
