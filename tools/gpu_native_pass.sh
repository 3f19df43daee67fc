#!/bin/bash
# A GPU call that fits into a few minutes of box time: everything goes through tools/native_harness (no torch import).
#   1. the headline launch with its spot-row check, 2. the same launch on the other schedules / geometries (timing + check),
#   3. rocprofv3 kernel trace of the harness, 4. the PMC passes of tools/gpu_pmc.sh over the harness, traffic counters first.
# usage: gpurun --timeout 200 -- 'bash tools/gpu_native_pass.sh <tag>'
tag=${1:-r04zw}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
H=tools/native_harness
run() { name=$1; shift; timeout 40 $H "$@" > $O/harness_$name.json 2> $O/harness_$name.err; echo "$name rc=$? $(cat $O/harness_$name.json | cut -c1-400)"; }
run hy720p --geom hy720p
run hy720p_v2 --geom hy720p --variant 2
run hy720p_pre --geom hy720p --prescaled
run wan720p --geom wan720p
run hy720p_f16 --geom hy720p --dtype f16
R=$PWD
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o $tag -- $R/$H --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom hy720p --check 0 > $R/$O/kt.log 2>&1)
timeout 20 python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/harness_kernel_trace.txt; head -5 $O/harness_kernel_trace.txt | cut -c1-170
PMC_CMD="$H --geom hy720p --warm 1 --reps 1 --check 0" PMC_PASS_TIMEOUT=40 PMC_ORDER="4 5 1 3 2 6" bash tools/gpu_pmc.sh $tag
