#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3_trace
cat > /tmp/c3_run.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch, bench
from tensorflowasr_amd import _lib
print(bench.extra_config3(_lib.lib(), torch.device("cuda:0"), with_cpu=False)["ms_per_step"])
PY
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c3_trace -- python /tmp/c3_run.py > gpurun_out/c3_trace/run.log 2>&1
tail -1 gpurun_out/c3_trace/run.log
python - <<'PY'
import csv, glob
rows = []
for p in glob.glob("gpurun_out/c3_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
st = [i for i, r in enumerate(rows) if "fft_stft" in r[2]]
print(len(rows), len(st))
for si in (10, 15):
    a, b = st[si], st[si + 1]
    seg = rows[a:b]
    span = rows[b][0] - seg[0][0]; busy = sum(e - s for s, e, _ in seg)
    print("step %d: %d kernels, span %.1f us, busy %.1f us, idle %.1f us" % (si, len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3))
    prev = seg[0]
    for r in seg[1:] + [rows[b]]:
        g = (r[0] - prev[1]) / 1e3
        if g > 2.5: print("  gap %.1f us before %s (after %s)" % (g, r[2].replace("(anonymous namespace)::", "")[:60], prev[2].replace("(anonymous namespace)::", "")[:45]))
        prev = r
    if si == 10:
        for s, e, k in seg: print("    %7.1f us  %s" % ((e - s) / 1e3, k.replace("(anonymous namespace)::", "").replace("void ", "")[:90]))
PY
