#!/bin/bash
O=gpurun_out/r03zd; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
SVG_FULL_GRID=1 OMP_NUM_THREADS=8 timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k full_reference_grid -n 12 > $O/varblock_fullgrid.txt 2>&1; echo "fullgrid rc=$?" >> $O/varblock_fullgrid.txt; tail -3 $O/varblock_fullgrid.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab","svg1_other_models"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
