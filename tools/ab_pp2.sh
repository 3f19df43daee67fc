#!/bin/bash
# same-box A/B of the default band schedule at HunyuanVideo 720p (kernel ms, HIP events; pre-scaled q as bench.py runs it):
# libsvgattn.so vs comparison builds lib/libsvgattn_<tag>.so given as arguments (python sparse-videogen_amd/build.py --tag <tag> with
# SVG_EXTRA_HIPCC_FLAGS="-DSVG_PP2_...")
for i in 1 2 3; do
  for t in cur "$@"; do
    [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
    SVG_ATTN_LIB=$PWD/sparse-videogen_amd/lib/$f timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu --no-profiler --no-dense --no-svg2 --no-step --no-ab 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['roofline']['kernel_ms'], d['roofline']['frac'], d['clock']['sclk_mhz_timed_steps'])"
  done
done
