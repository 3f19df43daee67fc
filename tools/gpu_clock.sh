#!/bin/bash
# effective shader clock of the attention kernel for a bench variant: GRBM_GUI_ACTIVE / kernel duration
export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/clk_$v; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out -o c -- python bench.py --steps 2 --warmup 1 --no-cpu --no-dense --no-profiler --heads spatial --variant $v > $out/log.txt 2>&1
  python - <<PY
import csv,glob
v="$v"
cc=[r for f in glob.glob("gpurun_out/clk_%s/**/*counter_collection.csv"%v,recursive=True) for r in csv.DictReader(open(f)) if "band_attn" in r["Kernel_Name"]]
kt=[r for f in glob.glob("gpurun_out/clk_%s/**/*kernel_trace.csv"%v,recursive=True) for r in csv.DictReader(open(f)) if "band_attn" in r["Kernel_Name"]]
g=sum(float(r["Counter_Value"]) for r in cc)/max(1,len(cc))
d=sum(float(r["End_Timestamp"])-float(r["Start_Timestamp"]) for r in kt)/max(1,len(kt))
print("variant %s: GRBM_GUI_ACTIVE=%.4g dur=%.3f ms -> clock %.3f GHz (if counter is summed over 8 XCDs)"%(v,g,d/1e6,g/8/d))
PY
done
