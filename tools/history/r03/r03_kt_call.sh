#!/bin/bash
# kernel trace of the (reduced) bench command on the final code: the rocprofv3 average of the headline kernel beside the bench line's own
R=$PWD; O=gpurun_out/r03zj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o r03zj -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step --no-ab > $R/$O/bench_under_rocprof.json 2>/dev/null
cd $R
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/bench_kernel_trace.txt; head -5 $O/bench_kernel_trace.txt | cut -c1-150; tail -c 600 $O/bench_under_rocprof.json
rm -rf $O/kt
