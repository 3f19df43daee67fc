#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_prescaled.py tests/test_gpu_processors.py tests/test_gpu_bench_contract.py tests/test_gpu_prologue.py -q -m gpu > $O/pytest_new.txt 2>&1; tail -25 $O/pytest_new.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "launch_order or notify" >> $O/pytest_new.txt 2>&1; tail -3 $O/pytest_new.txt
SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/libsvgattn_abl.so timeout 300 python tools/pp_trace.py 15616 0,3,8 > $O/pp_trace.txt 2>&1; cat $O/pp_trace.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab","prescaled_rel_l2_vs_default_kernel","dense_same_gpu","fp8_hy720p"):
    print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
