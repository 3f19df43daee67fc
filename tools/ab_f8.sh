#!/bin/bash
# same-box A/B of fp8 band kernels (kernel ms, HIP events, HunyuanVideo 720p): libsvgattn.so + comparison builds
# lib/libsvgattn_<tag>.so given as arguments
for i in 1 2 3; do
  for l in cur "$@"; do
    [ "$l" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$l.so
    SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/$f python bench.py --dtype fp8 --steps 8 --warmup 3 --no-cpu --no-profiler --no-dense --no-svg2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$l', d['roofline']['kernel_ms'], d['fp8']['rel_l2_vs_bf16_kernel_this_workload'])"
  done
done
