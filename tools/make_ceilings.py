"""tests/golden/parity_ceilings.json from the parity log of a GPU run of the test suite:

    MI355ASR_PARITY_LOG=gpurun_out/parity.jsonl MI355ASR_PARITY_CEILINGS=0 python -m pytest tests -m gpu -q      (on the GPU box)
    python tools/make_ceilings.py gpurun_out/parity.jsonl [more logs ...]                                          (here)

Every maxdiff() call of a GPU test logs {"tag": "<node id>#<n>", "max_abs_err": e} (tests/helpers.py).  The ceiling of a
comparison is FACTOR x the largest error any of the given logs recorded for it, never below FLOOR (bit-identical comparisons
record 0.0 and keep a ceiling of FLOOR: they are asserted with array_equal where identity is the claim)."""
import json
import os
import sys

FACTOR, FLOOR = 4.0, 1e-7
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(paths):
    worst, runs = {}, 0
    for p in paths:
        runs += 1
        for line in open(p):
            line = line.strip()
            if not line:
                continue
            e = json.loads(line)
            if "max_abs_err" in e and e.get("tag") and e["max_abs_err"] == e["max_abs_err"]:
                worst[e["tag"]] = max(worst.get(e["tag"], 0.0), float(e["max_abs_err"]))
    out = {"what": "regression ceilings of the GPU parity comparisons: %g x the recorded error, floor %g (tests/helpers.py)" % (FACTOR, FLOOR),
           "recorded_from": [os.path.basename(p) for p in paths], "comparisons": len(worst),
           "ceilings": {k: max(FACTOR * v, FLOOR) for k, v in sorted(worst.items())}}
    dst = os.path.join(ROOT, "tests", "golden", "parity_ceilings.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("wrote %s: %d comparisons from %d log(s); largest recorded error %.3g" % (dst, len(worst), runs, max(worst.values(), default=0.0)))


if __name__ == "__main__":
    main(sys.argv[1:])
