#!/bin/bash
# Round 5, GPU call zw: k-means centroid update walking clusters per workgroup with prefetched indices: groups per head sweep against the previous
# kernel (one workgroup per cluster); checksums of the SVG2 layer-call must not change
tag=${1:-r05zw}; O=gpurun_out/$tag; mkdir -p $O
run() { l=$1; g=$2; [ "$g" = "-" ] && unset SVG_KMEANS_UPDATE_GROUPS || export SVG_KMEANS_UPDATE_GROUPS=$g
  timeout 120 tools/native_svg2 --geom wan720p --two-streams --lib sparse-videogen_amd/lib/$l.so > $O/svg2_${l}_$g.json 2> $O/svg2_${l}_$g.err; echo "$l groups=$g rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_${l}_$g.json')); print(d['kmeans_init_50it_ms'], d['ms'], d['o_checksum'], d['map_checksum'])")"; }
run libsvgattn_kmold -
run libsvgattn -
for g in ${GROUPS_LIST:-16 26 32 52 100 1000}; do run libsvgattn_kmg $g; done 2>&1
run libsvgattn_kmold -
