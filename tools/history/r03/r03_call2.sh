#!/bin/bash
O=gpurun_out/r03b; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/libsvgattn_abl.so timeout 300 python tools/pp_trace.py 15616 0,3,8 > $O/pp_trace.txt 2>&1; cat $O/pp_trace.txt
for v in 3 6; do bash tools/gpu_pmc_svg2.sh r03b_v$v $v > $O/pmc_svg2_v$v.json 2>$O/pmc_svg2_v$v.err; cat $O/pmc_svg2_v$v.json | head -40; done
