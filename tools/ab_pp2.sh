#!/bin/bash
# same-box A/B of the band schedules at HunyuanVideo 720p (kernel ms, HIP events): libsvgattn.so variant 2 (two-phase, max-free
# softmax) vs variant 3 (one wave per SIMD), plus optional comparison builds lib/libsvgattn_<tag>.so given as arguments
for i in 1 2 3; do
  for spec in cur:2 cur:3 "$@"; do
    l=${spec%%:*}; v=${spec#*:}; [ "$v" = "$spec" ] && v=2
    [ "$l" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$l.so
    SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/$f python bench.py --steps 8 --warmup 3 --no-cpu --no-profiler --no-dense --no-svg2 --no-step --variant $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l v$v', d['roofline']['kernel_ms'])"
  done
done
