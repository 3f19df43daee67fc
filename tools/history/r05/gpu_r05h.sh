#!/bin/bash
# Round 5, GPU call h: block map second form (fp64 MFMA scores + in-wave sort) against the first form: map checksum, output checksum, stage time
tag=${1:-r05h}; O=gpurun_out/$tag; mkdir -p $O
for g in small wan720p; do for l in libsvgattn libsvgattn_dyn4w libsvgattn_dynold; do timeout 120 tools/native_svg2 --geom $g --two-streams --lib sparse-videogen_amd/lib/$l.so > $O/svg2_${g}_$l.json 2> $O/svg2_${g}_$l.err; echo "$g $l rc=$? $(python3 -c "
import json; d=json.load(open('$O/svg2_${g}_$l.json')); print(d['ms'], d['rel_l2'], d['o_checksum'], d['map_checksum'], d['density'])") $(tail -1 $O/svg2_${g}_$l.err)"; done; done
