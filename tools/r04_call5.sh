#!/bin/bash
# Round 4, GPU call 5: per-phase cycle trace of the 16x16x32 body beside the 32x32x16 ones; LDS counters of both (bank conflicts of the new V read pattern)
O=gpurun_out/r04e; mkdir -p $O
SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/libsvgattn_abl.so timeout 200 python tools/pp_trace.py 15616 0,3,9 2>&1 | grep -v amdgpu.ids | tee $O/pp_trace.txt
export TMPDIR=/tmp
for var in 8 2; do
  timeout 170 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc_v$var -o p -- python tools/one_launch.py plain $var > $O/pmc_v$var.log 2>&1
  python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$O/pmc_v$var/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "band_attn" in row.get("Kernel_Name",""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("variant $var:", {k: f"{sum(v)/len(v):.4g}" for k,v in sorted(agg.items())})
PY
done 2>&1 | tee $O/pmc_lds.txt
