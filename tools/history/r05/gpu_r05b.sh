#!/bin/bash
# Round 5, GPU call b (torch-free): energy-table kernel-like rows (second form), head_dim 64 on the 16x16x32 body
tag=${1:-r05b}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
echo "== energy table"
timeout 200 tools/energy_table 77 40000 kernel 2>&1 | tee $O/energy_table_kernel_rows.txt
echo "== head_dim 64: shipped four-waves 32x32x16 body (variant 0) vs the 16x16x32 body (variant 8)"
bash tools/gpu_native_ab.sh ${tag}_d64_v0 "--geom cog15 --check 6" lib/libsvgattn.so
bash tools/gpu_native_ab.sh ${tag}_d64_v8 "--geom cog15 --variant 8 --check 6" lib/libsvgattn.so lib/libsvgattn_d64pf2.so lib/libsvgattn_msum0.so lib/libsvgattn_msum0pf2.so
for d in ${tag}_d64_v0 ${tag}_d64_v8; do for f in gpurun_out/$d/*.json; do python3 -c "
import json,sys
d=json.load(open('$f'))
print('%-44s ms %.3f sclk %.1f mcycles %.2f frac %.4f rel_l2 %.3e'%('$f'.split('/',1)[1],d['ms_mean'],d['sclk_mhz'],d['mcycles'],d['frac_of_2500'],d['rel_l2']))"; done; done
cat gpurun_out/${tag}_d64_v8/*.err | sort | uniq -c | head
