#!/bin/bash
# Round 4, GPU call 2: energy table (fixed duty accounting, deeper DMA prefetch, MFMA shapes), A/B of the cyclic sweep start, diagnosis of the
# one red fuzz case, the whole GPU suite on the cleaned-up library (parked tests promoted, fuzz in the default run), the default bench line.
O=gpurun_out/r04b; mkdir -p $O
timeout 120 tools/energy_table 77 40000 2>&1 | tee $O/energy_table.txt
timeout 300 python tools/ab_rotate.py 4 2>&1 | grep -v amdgpu.ids | tee $O/ab_rotate.txt
timeout 120 python tools/diag_kmeans_fuzz.py 6 2>&1 | grep -v amdgpu.ids | tee $O/diag_kmeans_fuzz.txt
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/pytest_gpu.txt
timeout 300 python -m pytest tests/test_gpu_prescaled.py -q -s -k large_logits 2>&1 | grep -a "large logits\]\|passed\|failed" | tee $O/pytest_large_logits.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
