#!/bin/bash
# Round 5, GPU call zq: SVG2 layer-call on the torch-free driver, library builds A/B (ms of the phases, attention TFLOP/s, output checksum)
tag=${1:-r05zq}; O=gpurun_out/$tag; mkdir -p $O
for r in 1 2; do for l in ${LIBS:-libsvgattn libsvgattn_noidx}; do timeout 120 tools/native_svg2 --geom wan720p --two-streams --lib sparse-videogen_amd/lib/$l.so ${ARGS:-} > $O/svg2_${l}_$r.json 2> $O/svg2_${l}_$r.err; echo "$l $r rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_${l}_$r.json')); print(d['ms'], d['attention_tflops'], d['rel_l2'], d['o_checksum'])")"; done; done 2>&1 | tee $O/ab.txt
