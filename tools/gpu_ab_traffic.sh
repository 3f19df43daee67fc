#!/bin/bash
# Same-box A/B of libsvgattn builds on the headline launch (tools/native_harness --geom hy720p): kernel ms + granted clock (two interleaved rounds),
# then the two traffic counter passes (FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS) per library.
#   gpurun --timeout 600 -- 'bash tools/gpu_ab_traffic.sh <tag> cur noswz nomsum'        ("cur" = lib/libsvgattn.so, x = lib/libsvgattn_x.so)
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
H=tools/native_harness
lib() { [ "$1" = cur ] && echo $PWD/sparse-videogen_amd/lib/libsvgattn.so || echo $PWD/sparse-videogen_amd/lib/libsvgattn_$1.so; }
for round in 1 2; do for l in "$@"; do
  timeout 60 $H --lib $(lib $l) --geom hy720p --warm 2 --reps 5 > $O/time_${l}_$round.json 2> $O/time_${l}_$round.err
  echo "$l round $round: $(python3 -c "import json,sys; d=json.loads(open('$O/time_${l}_$round.json').read().strip().splitlines()[-1]); print({k: d.get(k) for k in ('ms_mean','sclk_mhz','mcycles','tflops','frac_of_2500','rel_l2','o_checksum') if k in d})" 2>/dev/null || head -c 300 $O/time_${l}_$round.json)"
done; done
for l in "$@"; do
  PMC_CMD="$H --lib $(lib $l) --geom hy720p --warm 1 --reps 1 --check 0" PMC_PASS_TIMEOUT=60 PMC_ORDER="4 5" bash tools/gpu_pmc.sh ${tag}_$l > $O/pmc_$l.txt 2>&1
  python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_$l/summary.txt ${tag}_$l "" "tools/gpu_ab_traffic.sh $tag: library $l" > $O/traffic_$l.json 2>> $O/pmc_$l.txt
  python3 -c "import json; d=json.load(open('$O/traffic_$l.json')); print('$l', 'traffic GB', round(d['traffic_bytes_per_launch']/1e9,2), 'l2 hit', round(d.get('l2_hit_rate') or 0,4), 'write MB', round((d.get('WRITE_SIZE_KB') or 0)/1e3,1))"
done
