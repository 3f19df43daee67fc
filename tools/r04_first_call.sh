#!/bin/bash
# Round 4, first GPU call (after tools/r04_prepare.sh): measure the three prepared experiments against the product build on one box.
O=gpurun_out/r04a; mkdir -p $O
L=$PWD/sparse-videogen_amd/lib
# 1. k-means V2: labels / counts / centroids must be bit-identical; time of the loop
timeout 300 python tools/ab_bitexact.py $L/libsvgattn.so $L/libsvgattn_km2.so --kmeans 2>&1 | grep -v amdgpu.ids | grep "k-means\|MISMATCH\|IDENTICAL" | tee $O/ab_kmeans_v2.txt
# 2. fp8 SVG2 with the MFMA row sum: error against the 16-bit kernel and time (bench_svg2 prints both)
for t in cur rs8 cur rs8; do
  [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
  SVG_ATTN_LIB=$L/$f timeout 200 python bench_svg2.py --fp8 --steps 4 --warmup 2 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 fp8 $t', d['ms'], 'rel_l2 vs 16-bit', d.get('rel_l2_vs_16bit_kernel'), 'spot rows', d.get('spot_rows_rel_l2_vs_torch_fp32'))"
done 2>&1 | tee $O/ab_svg2_fp8_rowsum.txt
for t in cur rs8 cur rs8; do   # ... and the SVG1 band kernel in fp8 (the same switch covers its two-phase body)
  [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
  SVG_ATTN_LIB=$L/$f timeout 200 python bench.py --dtype fp8 --steps 6 --warmup 2 --no-cpu --no-dense --no-svg2 --no-step --no-ab 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('band fp8 $t', d['roofline']['kernel_ms'], d['fp8'].get('rel_l2_vs_bf16_kernel_this_workload'))"
done 2>&1 | tee $O/ab_band_fp8_rowsum.txt
SVG_ATTN_LIB=$L/libsvgattn_rs8.so timeout 300 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_fullsize_svg2.py -q -k "fp8" 2>&1 | tail -3 | tee $O/pytest_fp8_rowsum.txt
# 3. head_dim 64 with the MFMA row sum: CogVideoX geometries, then the head_dim-64 parity tests on that build
for t in cur ms64 cur ms64; do
  [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
  echo "== $t"; SVG_ATTN_LIB=$L/$f timeout 200 python tools/svg1_models.py pre 2>>$O/err.txt | grep -i "cog"
done 2>&1 | tee $O/ab_cog_mfmasum.txt
SVG_ATTN_LIB=$L/libsvgattn_ms64.so timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prescaled.py tests/test_gpu_fullsize.py -q -k "64 or cog" 2>&1 | tail -3 | tee $O/pytest_d64_mfmasum.txt
