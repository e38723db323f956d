"""Appends the un-profiled step of the same box / same script run to a rocprofv3 summary (tools/profile_round.sh):
python tools/reconcile_profile.py BENCH_LINE.json >> profiles/<tag>_summary.md"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ev = sum(v["avg_ms"] * v["launches_per_step"] for v in d["kernels"].values())
r = d["region_ms_per_step"]
print()
print("Un-profiled, same box, same script run: **ms_per_step %.3f** over %d steps (%.2f s timed; K-step regions min / median / max "
      "%.3f / %.3f / %.3f ms per step); HIP-event kernel sum of one step %.3f ms (ms_per_step_with_kernel_events %.3f)."
      % (d["ms_per_step"], d["timed_steps_total"], d["timed_region_s"], r["min"], r["median"], r["max"], ev,
         d["ms_per_step_with_kernel_events"]))
