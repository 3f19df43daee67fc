#!/bin/bash
O=gpurun_out/r03zc; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -12 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab","svg1_other_models"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
