#!/bin/bash
# Round 4, GPU call 8: the whole GPU suite with the 16x16x32 body as the default at head_dim 128, then the default bench line
O=gpurun_out/r04h; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p",{}).get("attention_frac_of_2500tflops_bf16"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
