#!/bin/bash
# Round 5, GPU call c (torch-free): energy-table kernel-like rows on N(0,1) operand data; the row-sum-on-MFMA default against the round-4 body on
# Wan SVG1 and on the SVG2 layer-call; harness matrix of spot rows on the new default.
tag=${1:-r05c}; O=gpurun_out/$tag; mkdir -p $O
export TMPDIR=/tmp
echo "== energy table"
timeout 300 tools/energy_table 77 40000 kernel 2>&1 | tee $O/energy_table_kernel_rows.txt
echo "== A/B wan720p SVG1"
bash tools/gpu_native_ab.sh ${tag}_wan "--geom wan720p --check 6" lib/libsvgattn.so lib/libsvgattn_nomsum.so
echo "== A/B SVG2 wan720p"
for r in 1 2; do for l in libsvgattn libsvgattn_nomsum; do timeout 120 tools/native_svg2 --geom wan720p --lib sparse-videogen_amd/lib/$l.so > $O/svg2_${l}_$r.json 2> $O/svg2_${l}_$r.err; echo "$l $r rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_${l}_$r.json')); print(d['ms'], d['rel_l2'], d['o_checksum'])")"; done; done
echo "== matrix"
bash tools/gpu_native_matrix.sh ${tag}_matrix
