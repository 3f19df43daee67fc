#!/bin/bash
# Round 5, GPU call ze: profile16_kernel, finer phases of wave 0 (full kernel and golden output only)
tag=${1:-r05ze}; O=gpurun_out/$tag; mkdir -p $O
for l in ${LIBS:-libsvgattn_p16trace libsvgattn_p16trace7}; do
  timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/$l.so --geom hy720p --profiler --reps 10 > $O/prof_$l.json 2> $O/prof_$l.err
  echo "$l rc=$? $(python3 -c "
import json; d=json.load(open('$O/prof_$l.json')); print(d['ms_mean'], d['mse_sum'])")"; grep "prof phases" $O/prof_$l.err
done 2>&1 | tee $O/ab.txt
