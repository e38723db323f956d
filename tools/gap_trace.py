"""GPU idle time between the kernels of a loop, from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python tools/batch_sweep.py 1
    python tools/gap_trace.py gpurun_out/gap/**/*_kernel_trace.csv [last N kernels]

Prints, over the last N dispatches (default: the second half of the trace = the timed loop), the span from the first start to the
last end, the sum of kernel durations and the difference (dispatch gaps + host starvation), plus the largest gaps by successor."""
import csv
import glob
import sys
from collections import defaultdict

paths = [p for a in sys.argv[1:] if not a.isdigit() for p in glob.glob(a, recursive=True)]
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else len(rows) // 2
rows = rows[-n:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
print("dispatches %d  span %.1f us  kernel sum %.1f us  idle %.1f us (%.1f %%)" % (len(rows), span / 1e3, busy / 1e3, (span - busy) / 1e3,
                                                                                 100.0 * (span - busy) / span))
gaps = defaultdict(lambda: [0, 0])
for (s0, e0, _), (s1, e1, k1) in zip(rows, rows[1:]):
    g = gaps[k1.split("(")[0][-60:]]
    g[0] += max(0, s1 - e0)
    g[1] += 1
for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  %8.2f us avg gap in front of %4d x %s" % (t / c / 1e3, c, k))
