#!/bin/bash
# Round 5 validation pass on a GPU box (python): the GPU suite, the default bench line, the reference's complete varblock grid, then the torch-free
# harness pass (timing, kernel trace, PMC) on the same box.     gpurun --timeout 2700 -- 'bash tools/history/r05/gpu_r05n.sh <tag>'
tag=${1:-r05n}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" >> $O/pytest_gpu.txt; tail -5 $O/pytest_gpu.txt
t1=$(date +%s)
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s) - t1 )) s"; tail -3 $O/bench.err
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
try:
    d = json.loads(open(f"{O}/bench.json").read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line", e); sys.exit(0)
for k in ("value", "ms_per_step", "roofline", "clock", "same_box_ab", "output_checksum"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p", {}).get("ms"), d.get("svg2_wan720p", {}).get("kmeans_init_50it_ms"), "fp8", d.get("svg2_wan720p_fp8", {}).get("ms"))
print("step", {k: v for k, v in d.get("denoise_step_hy720p", {}).items() if "per_s" in k})
for n, r in (d.get("hbm_kernels", {}).get("kernels") or {}).items(): print("hbm", n, r["ms"], r["GBs"], r["frac_of_8TBs"])
print("hbm copy", d.get("hbm_kernels", {}).get("torch_copy_this_box"), d.get("hbm_kernels", {}).get("error"))
PY
t2=$(date +%s)
SVG_FULL_GRID=1 OMP_NUM_THREADS=8 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k full_reference_grid -n 12 > $O/varblock_fullgrid.txt 2>&1; echo "fullgrid rc=$? $(( $(date +%s) - t2 )) s" >> $O/varblock_fullgrid.txt; tail -3 $O/varblock_fullgrid.txt
bash tools/gpu_native_pass.sh ${tag} 2>&1 | tail -25
