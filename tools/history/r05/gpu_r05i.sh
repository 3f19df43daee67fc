#!/bin/bash
# Round 5, GPU call i: PMC passes over the SVG2 layer-call through native_svg2 (k-means assign / update, block map, attention)
tag=${1:-r05i}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/p$i -o p -- $R/tools/native_svg2 --lib $R/sparse-videogen_amd/lib/libsvgattn.so --geom wan720p --check 0 --warm 0 --reps 1 > $R/$O/p$i.log 2>&1)
  echo "pass $i rc=$?"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"{O}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        short = k.split("(")[0].replace("_ZN3svg", "")[:34]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"{O}/p*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        short = k.split("(")[0].replace("_ZN3svg", "")[:34]
        dur[short].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
out = {}
for k, d in agg.items():
    if not any(x in k for x in ("kmeans_assign", "kmeans_update", "dynmap", "varblock_attn", "sort_scatter")): continue
    row = {c: sum(v) / len(v) for c, v in d.items()}
    row["launches_seen"] = max(len(v) for v in d.values())
    row["ms_avg"] = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
    out[k] = row
    print(k, json.dumps(row))
json.dump(out, open(f"{O}/pmc_svg2_native.json", "w"), indent=1)
PY
