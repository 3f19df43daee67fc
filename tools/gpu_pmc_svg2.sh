#!/bin/bash
# L2 / fabric traffic of the SVG2 attention kernel (Wan 2.1 720p, bench_svg2.py): separate rocprofv3 --pmc passes (gpurun refuses
# --pmc combined with sys / hip traces), then profiles-ready JSON.   usage: tools/gpu_pmc_svg2.sh <tag> <variant> [--fp8]
tag=$1; variant=${2:--1}; shift; shift
export TMPDIR=/tmp
out=gpurun_out/pmc_svg2_$tag
mkdir -p $out
B="python bench_svg2.py --steps 1 --warmup 1 --variant $variant $@"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -o p$i -- $B > $out/p$i.log 2>&1
  echo "pass $i rc=$? : $set" >> $out/passes.txt
done
python - <<PY
import csv, glob, collections, json
out="$out"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")
        if "varblock_attn" not in k: continue
        agg[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out+"/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","?")
        if "varblock" in k: dur[k[:48]].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e6)
res={}
for k,d in agg.items():
    g=lambda c: (sum(d[c])/len(d[c])) if d.get(c) else None
    fetch, wr = g("FETCH_SIZE"), g("WRITE_SIZE")
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    res[k]={"kernel": k, "workload": "bench_svg2.py wan720p --variant $variant $@ (H=40 S=75600 D=128 QC=300 KC=1000), one launch",
            "source": "tools/gpu_pmc_svg2.sh $tag: separate rocprofv3 --kernel-trace --pmc passes, average over the launches seen",
            "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": wr,
            "traffic_bytes_per_launch": ((2.0*(fetch or 0.0))+(wr or 0.0))*1024.0,
            "TCC_HIT_sum": hit, "TCC_MISS_sum": miss, "TCC_REQ_sum": g("TCC_REQ_sum"),
            "l2_hit_rate": (hit/(hit+miss)) if hit is not None and miss else None,
            "kernel_ms_under_rocprof": (sum(dur[k])/len(dur[k])) if dur.get(k) else None,
            "note": "FETCH_SIZE x2: gfx950 correction (MI355X_MICROARCH.md); the counter sits between L2 and the fabric (Infinity Cache hits included)"}
others={k: round(sum(v)/len(v),4) for k,v in dur.items() if "varblock_attn" not in k}
print(json.dumps({"attention": list(res.values()), "other_varblock_kernels_ms": others}, indent=1))
PY
