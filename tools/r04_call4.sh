#!/bin/bash
# Round 4, GPU call 4: first run of the 16x16x32 two-phase body (csrc/attn_m16.h): parity tests, A/B on the headline workload, SVG2 layer-call
O=gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_m16.py -q -x 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/pytest_m16.txt
timeout 300 python tools/ab_m16.py 4 2 2>&1 | grep -v amdgpu.ids | tee $O/ab_m16.txt
for r in 1 2; do
  for var in -1 8; do
    timeout 200 python bench_svg2.py --variant $var --steps 4 --warmup 2 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 variant $var', d['ms'], 'tflops', d['attention_tflops_algorithmic'], 'spot rows', d.get('spot_rows_rel_l2_vs_torch_fp32'))"
  done
done 2>&1 | tee $O/ab_svg2_m16.txt
tail -3 $O/err.txt
