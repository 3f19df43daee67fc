#!/bin/bash
# Round 5, GPU call e: energy table with a DMA source that toggles (N(0,1) bf16) — kernel-like rows, then the whole table; the real kernel beside it
tag=${1:-r05e}; O=gpurun_out/$tag; mkdir -p $O
echo "== kernel-like rows, random DMA source"
timeout 300 tools/energy_table 77 40000 kernel 2>&1 | tee $O/energy_table_kernel_rows.txt
echo "== the real kernel beside it (same box)"
for f in normal zero; do timeout 60 tools/native_harness --geom hy720p --fill $f --check 0 > $O/fill_$f.json; python3 -c "
import json; d=json.load(open('$O/fill_$f.json')); print('$f', d['ms_mean'], d['sclk_mhz'], d['mcycles'], d['frac_of_2500'])"; done
echo "== kernel-like rows, constant DMA source (the flaw of rounds 4 - 5c)"
timeout 300 tools/energy_table 77 40000 kernel 1 2>&1 | tail -9 | tee $O/energy_table_kernel_rows_constant_src.txt
echo "== whole table, random DMA source"
timeout 400 tools/energy_table 77 40000 2>&1 | tee $O/energy_table_full.txt
