#!/bin/bash
# round 4, session b: all-64 fixture tests, adversarial operand bounds (two- vs three-term), bench line with the exact-product leg
O=gpurun_out/r04b; mkdir -p $O
export MI355ASR_PARITY_LOG=$PWD/$O/parity.jsonl
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -x -q -k "all_64" -s > $O/all64.log 2>&1; echo all64 rc=$?; grep -a "config 2, all" $O/all64.log; tail -3 $O/all64.log
timeout 300 python tools/adversarial_two_term.py > $O/adv_two.log 2>&1; echo adv2 rc=$?; grep CASE $O/adv_two.log
MI355ASR_PP=0 MI355ASR_PP_OUTGLU=0 MI355ASR_ATTN_TERMS=3 timeout 300 python tools/adversarial_two_term.py > $O/adv_three.log 2>&1; echo adv3 rc=$?; grep CASE $O/adv_three.log
timeout 600 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo bench rc=$?
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step", j["ms_per_step"], "value", j["value"], "roofline", j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["pipe"])
print("exact", j.get("exact_products"))
print({k:(v["avg_ms"], v["scheme"]) for k,v in j["kernels"].items()})
for c in ("config3","config5"):
    print(c, {k:j[c].get(k) for k in ("ms_per_step","ms_predict","ms_beam10","error")}, j[c].get("roofline"))
PY
