#!/bin/bash
# Round 5, GPU call f: where the block-map kernel's time is (ablation builds), kernel trace of the SVG2 layer-call through native_svg2
tag=${1:-r05f}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for l in libsvgattn libsvgattn_dyn1 libsvgattn_dyn2 libsvgattn_dyn4 libsvgattn_dyn7; do timeout 120 tools/native_svg2 --geom wan720p --check 0 --two-streams --lib sparse-videogen_amd/lib/$l.so > $O/svg2_$l.json 2> $O/svg2_$l.err; echo "$l rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_$l.json')); print(d['ms'])")"; done
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $R/tools/native_svg2 --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom wan720p --check 0 --reps 3 > $R/$O/kt.log 2>&1)
timeout 20 python3 tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/svg2_kernel_trace.txt; head -22 $O/svg2_kernel_trace.txt | cut -c1-190
