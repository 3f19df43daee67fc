#!/bin/bash
# Round 5 final pass on a GPU box: the GPU suite, the default bench line, the same command under rocprofv3 --kernel-trace --stats, the torch-free harness
# pass with the PMC sets, the SVG2 PMC passes.     gpurun --timeout 2700 -- 'bash tools/history/r05/gpu_r05z.sh <tag>'
tag=${1:-r05z}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
t0=$(date +%s)
timeout 1700 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
t1=$(date +%s)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t1 )) s"; tail -2 $O/bench.err
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
try:
    d = json.loads(open(f"{O}/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line", e); sys.exit(0)
for k in ("value", "ms_per_step", "roofline", "clock", "output_checksum"): print(k, d.get(k))
print("ab", {k: v for k, v in d.get("same_box_ab", {}).items() if k != "what"})
print("svg2", d.get("svg2_wan720p", {}).get("ms"), d.get("svg2_wan720p", {}).get("kmeans_init_50it_ms"), d.get("svg2_wan720p", {}).get("attention_sclk_mhz"), "fp8", d.get("svg2_wan720p_fp8", {}).get("ms"))
print("other", d.get("svg1_other_models"))
print("step", {k: v for k, v in d.get("denoise_step_hy720p", {}).items() if "per_s" in k})
for n, r in (d.get("hbm_kernels", {}).get("kernels") or {}).items(): print("hbm", n, r["ms"], r["GBs"], r["frac_of_8TBs"])
PY
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step --no-ab --no-hbm > $R/$O/bench_under_rocprof.json 2>/dev/null)
python3 tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/bench_kernel_trace.txt; head -6 $O/bench_kernel_trace.txt | cut -c1-170
bash tools/gpu_native_pass.sh ${tag} 2>&1 | head -8
bash tools/history/r05/gpu_r05i.sh ${tag}_svg2 2>&1 | grep "varblock_attn" | cut -c1-400
