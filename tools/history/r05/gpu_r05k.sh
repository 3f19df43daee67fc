#!/bin/bash
# Round 5, GPU call k: online profiler with the fast predicate on frame-major tiles against the previous build: ms, mse bits (must be equal), kernel trace
tag=${1:-r05k}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for g in hy720p wan720p cog15 hy480p; do for l in libsvgattn libsvgattn_profold; do timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/$l.so --geom $g --profiler --reps 10 > $O/prof_${g}_$l.json 2> $O/prof_${g}_$l.err; echo "$g $l rc=$? $(python3 -c "
import json; d=json.load(open('$O/prof_${g}_$l.json')); print(d['ms_mean'], d['gbps'], d['mse_sum'], d['mse_bits'])")"; done; done
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d $R/$O/kt_prof -o kt -- $R/tools/native_harness --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom hy720p --profiler --reps 5 > $R/$O/kt_prof.log 2>&1)
timeout 20 python3 tools/rocprof_summary.py $(find $O/kt_prof -name "*.db" | head -1) $O/profiler_kernel_trace.txt; head -6 $O/profiler_kernel_trace.txt | cut -c1-150
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_prof -o p -- $R/tools/native_harness --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom hy720p --profiler --warm 1 --reps 1 > $R/$O/pmc_prof.log 2>&1)
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_prof/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "profile_attn" in k: agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
