"""Per-kernel PMC averages from one or more `rocprofv3 --pmc ...` passes (each pass in its own directory).

  python tools/pmc_probe.py <dir> [<dir> ...]      -> table: kernel category x counter (average per launch)

Used for the stall analysis in DESIGN.md section 2 (SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_*,
SQ_VALU_MFMA_BUSY_CYCLES ...).  Counters are summed over all SEs/XCDs by rocprofv3; SQ_*_CYCLES counters
count per-wave (SQ_WAVE_CYCLES, SQ_WAIT_*) or per-SQ (SQ_BUSY_CYCLES) quad-cycles as the hardware defines them,
so read them as ratios against each other, not as absolute time."""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import category  # noqa: E402


def main():
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = category(r["Kernel_Name"])
                    tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[k][r["Counter_Name"]] += 1
    counters = sorted({c for k in tot for c in tot[k]})
    print("kernel," + ",".join(counters))
    for k in sorted(tot):
        print(k + "," + ",".join("%.4g" % (tot[k][c] / cnt[k][c]) if cnt[k][c] else "" for c in counters))


if __name__ == "__main__":
    main()
