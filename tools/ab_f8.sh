#!/bin/bash
# same-box A/B of the fp8 band kernel at HunyuanVideo 720p (bench.py --dtype fp8: kernel ms without the pre-pass): libsvgattn.so vs
# comparison builds lib/libsvgattn_<tag>.so given as arguments
for i in 1 2 3; do
  for t in cur "$@"; do
    [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
    SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/$f timeout 300 python bench.py --dtype fp8 --steps 6 --warmup 2 --no-cpu --no-profiler --no-dense --no-svg2 --no-step 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['roofline']['kernel_ms'], d['roofline']['frac'], d['fp8']['rel_l2_vs_bf16_kernel_this_workload'], (d.get('clock') or {}).get('sclk_mhz_timed_steps'))"
  done
done
