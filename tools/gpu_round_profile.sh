#!/bin/bash
# One GPU-box pass that produces everything the round's profiles/ entries come from (run through gpurun):
#   tools/gpu_round_profile.sh <tag>       e.g.  gpurun --timeout 2400 -- 'bash tools/gpu_round_profile.sh r04z'
# Outputs under gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>_*.
tag=${1:-r04z}
R=$GRAFT_REPO_ROOT
O=gpurun_out/$tag
mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step --no-ab > $R/$O/bench_under_rocprof.json 2>/dev/null
cd $R
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/bench_kernel_trace.txt
bash tools/gpu_pmc.sh $tag --no-svg2 --no-step > $O/pmc.log 2>&1
cp gpurun_out/pmc_$tag/summary.txt $O/pmc_summary.txt
python tools/pmc_traffic.py gpurun_out/pmc_$tag/summary.txt $tag > $O/pmc_traffic.json 2> $O/pmc_traffic.err
bash tools/gpu_pmc_svg2.sh $tag -1 > $O/pmc_traffic_svg2_raw.json 2> $O/pmc_svg2.err
timeout 600 python tools/svg1_models.py > $O/svg1_models.md 2>&1
timeout 300 python bench_svg2.py --workload hy720p > $O/svg2_hy720p.json 2> $O/svg2_hy.err
SVG_FULL_GRID=1 OMP_NUM_THREADS=8 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k full_reference_grid -n 12 > $O/varblock_fullgrid.txt 2>&1; echo "fullgrid rc=$?" >> $O/varblock_fullgrid.txt; tail -3 $O/varblock_fullgrid.txt
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
for k in ("value","ms_per_step","roofline","clock","same_box_ab"): print(k, d.get(k))
print("svg2", d.get("svg2_wan720p",{}).get("ms"), d.get("svg2_wan720p_fp8",{}).get("ms"))
print("step", {k:v for k,v in d.get("denoise_step_hy720p",{}).items() if "per_s" in k})
PY
head -4 $O/bench_kernel_trace.txt | cut -c1-180; cat $O/svg1_models.md | tail -12; cat $O/pmc_traffic.json | head -20
