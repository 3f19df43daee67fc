#!/bin/bash
# Production-size spot rows of every band schedule against the harness's own fp32 restatement (tools/native_harness.hip, C++, independent of the
# Python oracle): geometries x schedules x head placements x dtypes, one JSON line each.
#   gpurun --timeout 150 -- 'bash tools/gpu_native_matrix.sh <tag>'
tag=${1:-r04zu}; O=gpurun_out/$tag; mkdir -p $O; : > $O/spot_rows.jsonl; : > $O/rc.txt
H=tools/native_harness
run() { timeout 40 $H --warm 1 --reps 2 --check 12 "$@" >> $O/spot_rows.jsonl 2>> $O/err.txt; echo "rc=$? $*" >> $O/rc.txt; }
for g in hy720p wan720p hy480p; do
  for v in 0 1 2 3; do run --geom $g --variant $v --flags half; done
  run --geom $g --variant 0 --flags one
  run --geom $g --variant 0 --flags zero
  run --geom $g --variant 0 --dtype f16
  run --geom $g --variant 3 --dtype f16 --flags one
  run --geom $g --prescaled
  run --geom $g --prescaled --dtype f16
done
run --geom hy720p --variant 6
run --geom small --variant 0
run --geom small --variant 2 --dtype f16
cat $O/rc.txt | grep -v "rc=0" ; python3 - $O/spot_rows.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"{d['geom']:8s} {d['dtype']:4s} v{d['variant']} pre{d['prescaled']} flags {d['head_flags']:4s} {d['ms_mean']:8.3f} ms  frac {d['frac_of_2500']:.4f}  rows {d['spot_rows']}  rel_l2 {d['rel_l2']:.3e}  max_abs {d['max_abs']:.2e}")
PY
