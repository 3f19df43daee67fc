#!/bin/bash
# Round 5, GPU call zd: profile16_kernel after the uniform-branch predicate, packed rounding, lazy maximum, sorted rows: A/B against the first form on
# the four geometries, phases, ablations, the profiler tests
tag=${1:-r05zd}; O=gpurun_out/$tag; mkdir -p $O
for g in hy720p wan720p hy480p; do for l in libsvgattn libsvgattn_prof1; do timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/$l.so --geom $g --profiler --reps 10 > $O/prof_${g}_$l.json 2> $O/prof_${g}_$l.err; echo "$g $l rc=$? $(python3 -c "
import json; d=json.load(open('$O/prof_${g}_$l.json')); print(d['ms_mean'], d['gbps'], d['mse_sum'], d['mse_bits'])")"; done; done 2>&1 | tee $O/ab.txt
bash tools/history/r05/gpu_r05zb.sh $tag/abl | grep -v "chunk"
timeout 900 python -m pytest tests -m gpu -q -k "profil or sample_mse or mse or processor" -p no:cacheprovider 2>&1 | tail -5 | tee $O/pytest_profiler.txt
