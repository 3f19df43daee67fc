#!/bin/bash
# PMC counter passes for the dominant kernel (run on the GPU box through gpurun).  Each pass is its own rocprofv3 run
# with --kernel-trace only (gpurun refuses --pmc combined with sys/hip traces).  usage: tools/gpu_pmc.sh <tag> [bench args]
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
B="python bench.py --steps 1 --warmup 1 --no-cpu --no-dense --no-profiler --no-ab $@"
rocprofv3 -L > $out/counters_list.txt 2>&1
i=0
for set in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
 "FETCH_SIZE" \
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- $B > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set" >> $out/passes.txt
done
python - <<PY
import csv, glob, collections, os
out="$out"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")
        if "attn_" not in k or "profile" in k: continue
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out+"/summary.txt","w") as fo:
    for k,d in agg.items():
        fo.write(k+"\n")
        for c,v in sorted(d.items()):
            fo.write(f"   {c:32s} n={len(v)} mean={sum(v)/len(v):.6g}\n")
print(open(out+"/summary.txt").read())
PY
