#!/bin/bash
# same-box A/B of the SVG2 attention (bench_svg2.py, Wan 2.1 720p): libsvgattn.so + comparison builds lib/libsvgattn_<tag>.so
for i in 1 2; do
  for l in cur "$@"; do
    [ "$l" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$l.so
    SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/$f python bench_svg2.py --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', {k: d[k] for k in d if 'ms' in k or 'flop' in k.lower()})"
  done
done
