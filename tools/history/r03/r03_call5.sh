#!/bin/bash
O=gpurun_out/r03e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py -q -m gpu -x > $O/pytest_contract.txt 2>&1; tail -25 $O/pytest_contract.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab","prescaled_rel_l2_vs_default_kernel","dense_same_gpu"):
    print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k}, d.get("denoise_step_hy720p",{}).get("sparse_step"))
PY
