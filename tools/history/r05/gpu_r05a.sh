#!/bin/bash
# Round 5, first GPU call (torch-free, ~3 min of box time): what bounds the headline kernel, measured.
#   1. the real kernel on random / zero / constant inputs: ms, granted clock, Mcycles (the schedule-only ceiling)
#   2. same-box A/B of candidate bodies (row sum on the matrix pipe, static priority)
#   3. band sweep: tile-iterations/s, clock and L2 hit rate vs band width
#   4. energy table: the kernel-like rows
#   5. the online profiler through the harness (+ one PMC pass), native_svg2 first run
tag=${1:-r05a}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
H=tools/native_harness
R=$PWD
L=sparse-videogen_amd/lib
run() { name=$1; shift; timeout 60 $H "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$? $(cut -c1-600 $O/$name.json)"; }
echo "== 1. fills"
for f in normal zero const normal; do run fill_$f --geom hy720p --fill $f --check 4; done
run fill_normal_v2 --geom hy720p --variant 2 --check 0
run fill_zero_v2 --geom hy720p --variant 2 --fill zero --check 0
echo "== 2. A/B"
bash tools/gpu_native_ab.sh ${tag}_ab "--geom hy720p --check 6" lib/libsvgattn.so lib/libsvgattn_msum.so lib/libsvgattn_msum0.so lib/libsvgattn_prio3.so lib/libsvgattn_msumprio3.so
echo "== 3. band sweep"
for b in 512 1024 2048 4096 8192 15616 31232; do
  run band_$b --geom hy720p --band $b --check 0 --reps 3
  (cd /tmp && timeout 60 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/$O/pmc_band_$b -o p -- $R/$H --lib $R/$L/libsvgattn.so --geom hy720p --band $b --check 0 --warm 1 --reps 1 --no-clock > $R/$O/pmc_band_$b.log 2>&1)
done
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
for b in (512, 1024, 2048, 4096, 8192, 15616, 31232):
    try: d = json.loads(open(f"{O}/band_{b}.json").read())
    except Exception as e: print(b, "unreadable", e); continue
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{O}/pmc_band_{b}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "band_attn" in r.get("Kernel_Name", ""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    g = lambda c: sum(agg[c]) / len(agg[c]) if agg.get(c) else None
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    pairs = d["algorithmic_tflop"] * 1e12 / (4.0 * d["D"] * d["H"])
    iters = pairs / (256 * 64)
    print(json.dumps({"band": b, "ms": d["ms_mean"], "sclk_mhz": d["sclk_mhz"], "mcycles": d["mcycles"], "frac_of_2500": d["frac_of_2500"],
                      "tile_iterations_per_us": round(iters / (d["ms_mean"] * 1e3), 2), "cycles_per_tile_iteration_per_cu": round(d["mcycles"] * 1e6 * 256 / iters, 1),
                      "l2_hit_rate": round(hit / (hit + miss), 4) if hit and miss else None, "tcc_miss_per_launch": miss}))
PY
echo "== 4. energy table"
timeout 120 tools/energy_table 77 20000 kernel 2>&1 | tee $O/energy_table_kernel_rows.txt
echo "== 5. profiler, native_svg2"
run profiler_hy720p --geom hy720p --profiler --reps 10
run profiler_wan720p --geom wan720p --profiler --reps 10
(cd /tmp && timeout 60 rocprofv3 --kernel-trace --stats -d $R/$O/kt_prof -o kt -- $R/$H --lib $R/$L/libsvgattn.so --geom hy720p --profiler --reps 5 > $R/$O/kt_prof.log 2>&1)
timeout 20 python3 tools/rocprof_summary.py $(find $O/kt_prof -name "*.db" | head -1) $O/profiler_kernel_trace.txt; head -8 $O/profiler_kernel_trace.txt | cut -c1-170
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  (cd /tmp && timeout 60 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$O/pmc_prof_$n -o p -- $R/$H --lib $R/$L/libsvgattn.so --geom hy720p --profiler --warm 1 --reps 1 > $R/$O/pmc_prof_$n.log 2>&1)
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_prof_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "profile_" in k: agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: sum(v) / len(v) for c, v in d.items()})
PY
for g in small wan720p; do timeout 120 tools/native_svg2 --geom $g > $O/native_svg2_$g.json 2> $O/native_svg2_$g.err; echo "native_svg2 $g rc=$? $(cut -c1-900 $O/native_svg2_$g.json) $(tail -2 $O/native_svg2_$g.err)"; done
