#!/bin/bash
# Round 5, GPU call zp: online profiler, KV chunks per head (workgroups per CU) on the head_dim 64 geometries and the head_dim 128 ones
tag=${1:-r05zp}; O=gpurun_out/$tag; mkdir -p $O
for g in cog15 cog480p hy720p wan720p; do for n in 0 8 10 12 16 21 24 32; do
  [ $n = 0 ] && unset SVG_PROF_CHUNKS || export SVG_PROF_CHUNKS=$n
  timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/libsvgattn_chunks.so --geom $g --profiler --reps 10 > $O/p.json 2> $O/p.err; echo "$g chunks=$n rc=$? $(python3 -c "
import json; d=json.load(open('$O/p.json')); print(d['ms_mean'], d['mse_sum'])")"; done; done 2>&1 | tee $O/chunks.txt
