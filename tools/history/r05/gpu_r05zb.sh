#!/bin/bash
# Round 5, GPU call zb: profile16_kernel — per-phase ticks of wave 0 (-DSVG_PROF_TRACE) and timing-only ablations (SVG_P16_ABL: 1 no mask predicates,
# 2 no P V of the masked outputs, 4 no wait on the staged tile; results of those are wrong by construction)
tag=${1:-r05zb}; O=gpurun_out/$tag; mkdir -p $O
for l in libsvgattn libsvgattn_p16trace libsvgattn_p16abl1 libsvgattn_p16abl2 libsvgattn_p16abl4 libsvgattn_p16abl7; do
  timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/$l.so --geom hy720p --profiler --reps 10 > $O/prof_$l.json 2> $O/prof_$l.err
  echo "$l rc=$? $(python3 -c "
import json; d=json.load(open('$O/prof_$l.json')); print(d['ms_mean'], d['mse_sum'])")"; grep "prof phases\|prof trace\|chunk  [0-3]:" $O/prof_$l.err
done 2>&1 | tee $O/ab.txt
