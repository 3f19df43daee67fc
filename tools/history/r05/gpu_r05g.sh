#!/bin/bash
# Round 5, GPU call g: label sort with prefetched labels (native_svg2 checksum must not move), kernel trace
tag=${1:-r05g}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for m in "" "--two-streams"; do timeout 120 tools/native_svg2 --geom wan720p $m > $O/svg2_${m#--}.json 2> $O/svg2_${m#--}.err; echo "[$m] rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_${m#--}.json')); print(d['kmeans_init_50it_ms'], d['ms'], d['rel_l2'], d['o_checksum'])")"; done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $R/tools/native_svg2 --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom wan720p --check 0 --reps 3 > $R/$O/kt.log 2>&1)
timeout 20 python3 tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/svg2_kernel_trace.txt; head -12 $O/svg2_kernel_trace.txt | cut -c1-150
