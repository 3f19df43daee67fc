#!/bin/bash
# L2 hit rate / fabric traffic of the band kernel per cyclic-start setting (SVG_BAND_ROTATE), one rocprofv3 --pmc pass each.
# usage: tools/pmc_rotate.sh <outdir> [plain|pre] [settings, default "0 1 2 3"]
out=$1; mode=${2:-plain}; sets=${3:-"0 1 2 3"}
export TMPDIR=/tmp
mkdir -p $out
for r in $sets; do
  for set in "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -c1-5)
    SVG_BAND_ROTATE=$r timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/r${r}_$tag -o p -- python tools/one_launch.py $mode > $out/r${r}_$tag.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
out="$out"
for r in "$sets".split():
    agg=collections.defaultdict(list)
    for f in glob.glob(f"{out}/r{r}_*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "band_attn" in row.get("Kernel_Name",""):
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    m={k:sum(v)/len(v) for k,v in agg.items()}
    if not m: print("rotate",r,"no data"); continue
    hit=m.get("TCC_HIT_sum",0); miss=m.get("TCC_MISS_sum",0)
    traffic=(2*m.get("FETCH_SIZE",0)+m.get("WRITE_SIZE",0))*1024
    print(f"rotate {r} ($mode): L2 hit rate {hit/max(hit+miss,1):.4f}  fabric traffic {traffic/1e9:.2f} GB / launch  (FETCH_SIZE {m.get('FETCH_SIZE',0):.4g} KB x2, WRITE_SIZE {m.get('WRITE_SIZE',0):.4g} KB)  TCC_REQ {m.get('TCC_REQ_sum',0):.4g}  EA0_RDREQ {m.get('TCC_EA0_RDREQ_sum',0):.4g}")
PY
