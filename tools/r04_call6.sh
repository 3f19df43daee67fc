#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_m16.py -q -x -k "band_attention or fused or varblock_fused" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/pytest_m16.txt
SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/libsvgattn_abl.so timeout 200 python tools/pp_trace.py 15616 0,9 2>&1 | grep -v amdgpu.ids | tee $O/pp_trace.txt
timeout 300 python tools/ab_m16.py 4 2 2>&1 | grep -v amdgpu.ids | tee $O/ab_m16.txt
export TMPDIR=/tmp
for var in 8; do
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
