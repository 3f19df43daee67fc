#!/bin/bash
# Round 5, GPU call za: online profiler, second form (one score tile for the three outputs, profile16_kernel) against the first form built from the same tree
# (-DSVG_PROF_FIRST_FORM): ms, mse sums (a shared softmax reference: equal to float rounding, not bit-equal), then the profiler tests.
tag=${1:-r05za}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for g in hy720p wan720p cog15 hy480p; do for l in libsvgattn libsvgattn_prof1; do timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/$l.so --geom $g --profiler --reps 10 > $O/prof_${g}_$l.json 2> $O/prof_${g}_$l.err; echo "$g $l rc=$? $(python3 -c "
import json; d=json.load(open('$O/prof_${g}_$l.json')); print(d['ms_mean'], d['gbps'], d['mse_sum'], d['mse_bits'])")"; done; done 2>&1 | tee $O/ab.txt
timeout 900 python -m pytest tests -m gpu -q -k "profil or sample_mse or mse or processor" -p no:cacheprovider 2>&1 | tail -15 | tee $O/pytest_profiler.txt
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d $R/$O/kt_prof -o kt -- $R/tools/native_harness --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom hy720p --profiler --reps 5 > $R/$O/kt_prof.log 2>&1)
timeout 20 python3 tools/rocprof_summary.py $(find $O/kt_prof -name "*.db" | head -1) $O/profiler_kernel_trace.txt; head -6 $O/profiler_kernel_trace.txt | cut -c1-150
