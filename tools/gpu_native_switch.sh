#!/bin/bash
# svg_band_attention_switch through tools/native_harness: flag 0 (sparse mask + placement) against the plain call's output checksum, flag 1
# (dense alternative, placement ignored) against the harness's fp32 spot rows under the dense mask; every geometry given, bf16 and fp16.
#   gpurun --timeout 100 -- 'bash tools/gpu_native_switch.sh <tag> cog15 cog480p small64'
tag=$1; shift; O=gpurun_out/$tag; mkdir -p $O; : > $O/switch.jsonl
H=tools/native_harness
for g in "$@"; do for dt in bf16 f16; do for fl in half one; do
  timeout 40 $H --geom $g --dtype $dt --flags $fl --variant 0 --warm 1 --reps 2 >> $O/switch.jsonl 2>> $O/err.txt || echo "plain rc=$? $g $dt $fl"
  timeout 40 $H --geom $g --dtype $dt --flags $fl --switch 0 --warm 1 --reps 2 >> $O/switch.jsonl 2>> $O/err.txt || echo "switch0 rc=$? $g $dt $fl"
  timeout 40 $H --geom $g --dtype $dt --flags $fl --switch 1 --warm 1 --reps 2 >> $O/switch.jsonl 2>> $O/err.txt || echo "switch1 rc=$? $g $dt $fl"
done; done; done
python3 - $O/switch.jsonl <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for i in range(0, len(rows) - 2, 3):
    p, s0, s1 = rows[i:i + 3]
    print(f"{p['geom']:8s} {p['dtype']:4s} flags {p['head_flags']:4s}  plain {p['ms_mean']:8.3f} ms {p['rel_l2']:.2e}   switch(0) {s0['ms_mean']:8.3f} ms {s0['rel_l2']:.2e} "
          f"checksum {'EQUAL' if p['o_checksum'] == s0['o_checksum'] else 'differs'}   switch(1, dense) {s1['ms_mean']:8.3f} ms rel_l2 {s1['rel_l2']:.2e} rows {s1['spot_rows']}")
print("runs", len(rows))
PY
