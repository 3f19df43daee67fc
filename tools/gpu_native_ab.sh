#!/bin/bash
# Same-box A/B of library builds through tools/native_harness (seconds of GPU-box time): every build in turn, ROUNDS times, so that clock
# drift shows as spread inside a build's column and not as a difference between builds; o_checksum equal = bit-identical outputs.
#   gpurun --timeout 120 -- 'bash tools/gpu_native_ab.sh <tag> "<harness args>" lib/libsvgattn.so lib/libsvgattn_<x>.so ...'
tag=$1; args=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for r in 1 2 3; do
  for lib in "$@"; do
    n=$(basename $lib .so)
    timeout 40 tools/native_harness --lib sparse-videogen_amd/$lib $args > $O/${n}_$r.json 2> $O/${n}_$r.err || echo "$n round $r rc=$?"
  done
done
python3 - "$O" <<'PY'
import glob, json, os, sys, collections
rows = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        print(f, "unreadable", e); continue
    rows[os.path.basename(f).rsplit("_", 1)[0]].append(d)
for n, ds in rows.items():
    print(f"{n:28s} ms " + " ".join(f"{d['ms_mean']:.3f}" for d in ds) + f"   min launch {min(min(d['ms']) for d in ds):.3f}   rel_l2 {ds[0]['rel_l2']:.3e}   checksum " + ",".join(sorted({d['o_checksum'] for d in ds})))
PY
