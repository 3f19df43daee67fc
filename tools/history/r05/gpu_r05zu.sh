#!/bin/bash
# Round 5, closing pass (budget-trimmed): the GPU test files not covered by r05zt (processors, bench contract, native harness, varblock, ...), the default
# bench line, the same command under rocprofv3 --kernel-trace --stats.
tag=${1:-r05zu}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
t0=$(date +%s)
timeout 400 python -m pytest tests -q -m gpu --ignore=tests/test_gpu_triton_golden.py --ignore=tests/test_gpu_glue.py --ignore=tests/test_gpu_kernels.py -p no:cacheprovider > $O/pytest_gpu_rest.txt 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" >> $O/pytest_gpu_rest.txt; tail -3 $O/pytest_gpu_rest.txt
t1=$(date +%s)
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t1 )) s"
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
d = json.loads(open(f"{O}/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline", "online_profiler", "clock", "output_checksum"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p", {}).get("ms"))
print("step", {k: v for k, v in d.get("denoise_step_hy720p", {}).items() if "per_s" in k})
for n, r in (d.get("hbm_kernels", {}).get("kernels") or {}).items(): print("hbm", n, r["ms"], r["GBs"], r["frac_of_8TBs"])
PY
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step --no-ab --no-hbm > $R/$O/bench_under_rocprof.json 2>/dev/null)
python3 tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/bench_kernel_trace.txt; head -8 $O/bench_kernel_trace.txt | cut -c1-170
