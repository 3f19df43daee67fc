mkdir -p gpurun_out/m
python bench.py > gpurun_out/m/bench.json 2> gpurun_out/m/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/m/kt -o r01m -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu --no-dense > $GRAFT_REPO_ROOT/gpurun_out/m/bench_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $(find gpurun_out/m/kt -name "*.db" | head -1) gpurun_out/m/kernel_trace.txt
bash tools/gpu_pmc.sh m > gpurun_out/m/pmc.log 2>&1
cp gpurun_out/pmc_m/summary.txt gpurun_out/m/pmc_summary.txt
python tools/wg_timeline.py 15616 alt > gpurun_out/m/wg_timeline.txt 2>&1
python tools/svg1_models.py > gpurun_out/m/svg1_models.md 2>&1
python bench_svg2.py > gpurun_out/m/svg2_wan.json 2>gpurun_out/m/svg2_wan.err
python bench_svg2.py --workload hy720p > gpurun_out/m/svg2_hy.json 2>gpurun_out/m/svg2_hy.err
tail -c 400 gpurun_out/m/bench.json; head -4 gpurun_out/m/kernel_trace.txt | cut -c1-180; cat gpurun_out/m/svg1_models.md; tail -2 gpurun_out/m/svg2_wan.json
