#!/bin/bash
# Round 5, last GPU call: kernel trace of the SVG2 layer-call on the torch-free driver (per-launch times of the k-means kernels after the update change)
tag=${1:-r05zzz}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $R/tools/native_svg2 --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom wan720p --two-streams --check 0 --reps 3 > $R/$O/kt.log 2>&1)
timeout 20 python3 tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) $O/svg2_kernel_trace.txt; head -14 $O/svg2_kernel_trace.txt | cut -c1-160
