#!/bin/bash
# One GPU-box pass that produces everything the round's profiles/ entries come from (run through gpurun):
#   tools/gpu_round_profile.sh <tag>       e.g.  gpurun --timeout 1500 -- 'bash tools/gpu_round_profile.sh r02a'
# Outputs under gpurun_out/<tag>/ ; copy what is to be judged into profiles/<tag>_*.
tag=${1:-r02a}
R=$GRAFT_REPO_ROOT
O=gpurun_out/$tag
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/$O/kt -o $tag -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step > $R/$O/bench_under_rocprof.json 2>/dev/null
cd $R
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/bench_kernel_trace.txt
bash tools/gpu_pmc.sh $tag --no-svg2 --no-step > $O/pmc.log 2>&1
cp gpurun_out/pmc_$tag/summary.txt $O/pmc_summary.txt
python tools/pmc_traffic.py gpurun_out/pmc_$tag/summary.txt $tag > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python tools/svg1_models.py > $O/svg1_models.md 2>&1
python bench_svg2.py --workload hy720p > $O/svg2_hy720p.json 2> $O/svg2_hy.err
tail -c 1500 $O/bench.json; head -4 $O/bench_kernel_trace.txt | cut -c1-180; cat $O/svg1_models.md; cat $O/pmc_traffic.json
