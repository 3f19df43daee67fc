#!/bin/bash
# Two library builds side by side through tools/native_harness: the same configurations on both, spot rows against the harness's fp32
# restatement on each, and the output checksums compared — equal checksums = bit-identical outputs at production size.
#   gpurun --timeout 100 -- 'bash tools/gpu_native_bitexact.sh <tag> lib/libsvgattn_old.so lib/libsvgattn.so'
tag=$1; A=$2; B=$3; O=gpurun_out/$tag; mkdir -p $O; : > $O/a.jsonl; : > $O/b.jsonl
H=tools/native_harness
cfg() { timeout 40 $H --lib sparse-videogen_amd/$A --warm 1 --reps 3 "$@" >> $O/a.jsonl 2>> $O/err.txt || echo "A rc=$? $*"; timeout 40 $H --lib sparse-videogen_amd/$B --warm 1 --reps 3 "$@" >> $O/b.jsonl 2>> $O/err.txt || echo "B rc=$? $*"; }
for g in cog15 cog480p small64; do
  for fl in half one zero; do cfg --geom $g --variant 2 --flags $fl; done
  cfg --geom $g --variant 2 --dtype f16
  cfg --geom $g --variant 2 --dtype f16 --flags one
done
python3 - $O <<'PY'
import json, sys
a = [json.loads(l) for l in open(sys.argv[1] + "/a.jsonl")]; b = [json.loads(l) for l in open(sys.argv[1] + "/b.jsonl")]
bad = 0
for x, y in zip(a, b):
    same = x["o_checksum"] == y["o_checksum"]
    bad += not same
    print(f"{x['geom']:8s} {x['dtype']:4s} flags {x['head_flags']:4s}  A {x['ms_mean']:7.3f} ms  B {y['ms_mean']:7.3f} ms  ({100 * (y['ms_mean'] / x['ms_mean'] - 1):+5.1f} %)  rel_l2 {x['rel_l2']:.2e} / {y['rel_l2']:.2e}  checksum {'EQUAL' if same else 'DIFFERENT'} {y['o_checksum']}")
print("configurations", len(a), len(b), "different", bad)
PY
# the new default at head_dim 64 (variant 0) on B against the explicit two-phase schedule on A
timeout 40 $H --lib sparse-videogen_amd/$A --geom cog15 --variant 2 --warm 1 --reps 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('A variant 2', d['ms_mean'], d['o_checksum'])"
timeout 40 $H --lib sparse-videogen_amd/$B --geom cog15 --variant 0 --warm 1 --reps 3 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('B variant 0', d['ms_mean'], d['o_checksum'])"
