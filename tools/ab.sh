#!/bin/bash
# same-box A/B of attention libraries: tools/ab.sh <lib suffixes...>  ("" = sparse-videogen_amd/lib/libsvgattn.so)
for i in 1 2; do for l in "$@"; do
  [ "$l" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$l.so
  SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/$f python bench.py --steps 5 --warmup 2 --no-cpu --no-profiler --no-dense 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['ms_per_step'])"
done; done
