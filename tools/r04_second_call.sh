#!/bin/bash
# Round 4, second GPU call: first run of everything that was written after round 3's GPU budget was spent (all opt-in, none of it in the
# default `-m gpu` run).  gpurun --timeout 900 -- 'bash tools/r04_second_call.sh'
O=gpurun_out/r04b; mkdir -p $O
# 1. tests/test_gpu_experimental.py: the mixed-precision SVG2 body (csrc/attn_f8pv.h), the pre-scaled SVG2 body (svg_varblock_attention_pre, with
#    its timing print), the k-means halves, and the parity tests parked there: the product's Hunyuan / CogVideoX / Cosmos processor __call__,
#    Wan cross attention + I2V branch and Wan block forward against the reference's executed ones (once green they move to test_gpu_triton_golden.py)
SVG_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_experimental.py -q -s 2>&1 | grep -v amdgpu.ids | tail -60 | tee $O/pytest_experimental.txt
# 2. random-geometry fuzz of the HIP kernels against the oracle (tests/test_gpu_fuzz.py, opt-in): first run ever
SVG_FUZZ=25 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -x 2>&1 | tail -8 | tee $O/pytest_fuzz.txt
# 3. the pre-scaled SVG2 path in the layer-call bench (Wan 720p): default, --pre, default, --pre
for m in "" "--pre" "" "--pre"; do
  timeout 200 python bench_svg2.py $m --steps 4 --warmup 2 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 16-bit [$m]', d['ms'], 'spot rows', d.get('spot_rows_rel_l2_vs_torch_fp32'))"
done 2>&1 | tee $O/ab_svg2_pre.txt
