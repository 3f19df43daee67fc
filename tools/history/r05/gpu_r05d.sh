#!/bin/bash
# Round 5, GPU call d (torch-free): energy table third form; SVG2 k-means chains on two streams; profiler chunk count sweep
tag=${1:-r05d}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
echo "== energy table"
timeout 300 tools/energy_table 77 40000 kernel 2>&1 | tee $O/energy_table_kernel_rows.txt
echo "== the real kernel beside it (same box)"
for f in normal zero; do timeout 60 tools/native_harness --geom hy720p --fill $f --check 0 > $O/fill_$f.json; python3 -c "
import json; d=json.load(open('$O/fill_$f.json')); print('$f', d['ms_mean'], d['sclk_mhz'], d['mcycles'], d['frac_of_2500'])"; done
echo "== SVG2 two streams"
for r in 1 2; do for m in "" "--two-streams"; do timeout 120 tools/native_svg2 --geom wan720p $m > $O/svg2_${r}_${m#--}.json 2> $O/svg2_${r}_${m#--}.err; echo "[$m] $r rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_${r}_${m#--}.json')); print(d['kmeans_init_50it_ms'], d['ms'], d['rel_l2'], d['o_checksum'])")"; done; done
echo "== profiler chunks"
for c in 0 16 21 32 42 64; do SVG_PROF_CHUNKS=$c; [ $c = 0 ] && unset SVG_PROF_CHUNKS || export SVG_PROF_CHUNKS; timeout 60 tools/native_harness --lib sparse-videogen_amd/lib/libsvgattn_profenv.so --geom hy720p --profiler --reps 10 > $O/prof_chunks_$c.json 2> $O/prof_chunks_$c.err; echo "chunks $c rc=$? $(python3 -c "
import json; d=json.load(open('$O/prof_chunks_$c.json')); print(d['ms_mean'], d['mse_sum'], d['mse_bits'])")"; done
