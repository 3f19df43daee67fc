#!/bin/bash
# Short GPU pass on HEAD: the GPU suite, cycles / clock of every schedule, the default bench line.
tag=${1:-r04zx}; O=gpurun_out/$tag; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 400 python tools/clock_by_variant.py 6 2>&1 | grep -v amdgpu.ids | tee $O/clock_by_variant.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
