#!/bin/bash
# strided attention tensors: parity (bit-exact against the contiguous calls), then the A/B at the BASELINE geometries and both denoise steps
tag=${1:-r06s}; O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_strided.py tests/test_gpu_m16.py -q -m gpu -x > $O/pytest_strided.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_strided.txt
grep -v amdgpu.ids $O/pytest_strided.txt | tail -15
timeout 600 python tools/ab_strided.py both 6 2> $O/ab.err | tee $O/ab_strided.jsonl
timeout 900 python bench_step.py --model hy720p --steps 2 --warmup 1 > $O/step_hy.json 2> $O/step_hy.err; echo "step hy rc=$?"
timeout 900 python bench_step.py --model wan720p --steps 2 --warmup 1 > $O/step_wan.json 2> $O/step_wan.err; echo "step wan rc=$?"
python - <<PY
import json
for f in ("step_hy","step_wan"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d.get("denoise_steps_per_s"), d.get("sparse_step",{}).get("ms"), d.get("sparse_step",{}).get("step_breakdown_ms"), d.get("dense_step",{}).get("ms"))
    except Exception as e: print(f, "ERR", e)
PY
for f in $O/*.err; do tail -n 3 $f; done
