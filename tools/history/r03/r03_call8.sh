#!/bin/bash
L=$PWD/sparse-videogen_amd/lib
SVG_ATTN_LIB=$L/libsvgattn_ob.so timeout 600 python -m pytest tests/test_gpu_prescaled.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_svg2.py -q -m gpu -x 2>&1 | tail -3
SVG_ATTN_LIB=$L/libsvgattn_ob.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "band_attention or varblock_attention_reference_grid_sample or notify" 2>&1 | tail -3
bash tools/ab_pp2.sh ob
for i in 1 2; do for t in cur ob; do [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
  SVG_ATTN_LIB=$L/$f timeout 300 python bench_svg2.py --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 $t', d['ms'])"; done; done
