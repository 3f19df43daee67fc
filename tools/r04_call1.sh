#!/bin/bash
# Round 4, GPU call 1: the energy table, the parked tests (first run ever), the new contract / large-logit tests, and the A/B of the
# experiments prepared at the end of round 3 (tools/r04_prepare.sh built the tagged libraries).  gpurun --timeout 1500 -- 'bash tools/r04_call1.sh'
O=gpurun_out/r04a; mkdir -p $O
L=$PWD/sparse-videogen_amd/lib
# 0. energy table of the headline kernel's instruction mix (tools/energy_table.hip, built on the build machine)
timeout 120 tools/energy_table 77 40000 2>&1 | tee $O/energy_table.txt
# 1. parked tests: mixed-precision SVG2 body, pre-scaled SVG2 body (+ timing print), k-means halves, processors vs the executed reference
SVG_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_gpu_experimental.py -q -s -rA 2>&1 | grep -v amdgpu.ids | tail -80 | tee $O/pytest_experimental.txt
# 2. random-geometry fuzz (first run ever)
SVG_FUZZ=25 timeout 400 python -m pytest tests/test_gpu_fuzz.py -q 2>&1 | tail -15 | tee $O/pytest_fuzz.txt
# 3. new this round: bench contract (svg2 measure, extras, exit code) and pre-scaled q against the reference formulation on large logits
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -q -k "svg2_measure or extras" 2>&1 | tail -15 | tee $O/pytest_contract.txt
timeout 300 python -m pytest tests/test_gpu_prescaled.py -q -s -k large_logits 2>&1 | grep -v amdgpu.ids | tail -40 | tee $O/pytest_large_logits.txt
# 4. k-means V2: labels / counts / centroids must be bit-identical; time of the loop
timeout 300 python tools/ab_bitexact.py $L/libsvgattn.so $L/libsvgattn_km2.so --kmeans 2>&1 | grep -v amdgpu.ids | grep "k-means\|MISMATCH\|IDENTICAL" | tee $O/ab_kmeans_v2.txt
# 5. SVG2 layer-call (Wan 720p): default, --pre, fp8 (shipped), fp8 with the MFMA row sum; two rounds
for r in 1 2; do
  for m in "" "--pre" "--fp8" "--fp8:rs8" "--fp8pv" "--materialize"; do
    f=libsvgattn.so; arg=$m
    [ "$m" = "--fp8:rs8" ] && f=libsvgattn_rs8.so && arg="--fp8"
    SVG_ATTN_LIB=$L/$f timeout 200 python bench_svg2.py $arg --steps 4 --warmup 2 2>>$O/err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('svg2 [$m]', d['ms'], 'tflops', d['attention_tflops_algorithmic'], 'rel_l2 vs 16-bit', d.get('rel_l2_vs_16bit_kernel'), 'spot rows', d.get('spot_rows_rel_l2_vs_torch_fp32'))"
  done
done 2>&1 | tee $O/ab_svg2.txt
# 6. head_dim 64 with the MFMA row sum: CogVideoX geometries, then the head_dim-64 parity tests on that build
for t in cur ms64 cur ms64; do
  [ "$t" = "cur" ] && f=libsvgattn.so || f=libsvgattn_$t.so
  echo "== $t"; SVG_ATTN_LIB=$L/$f timeout 200 python tools/svg1_models.py pre 2>>$O/err.txt | grep -i "cog"
done 2>&1 | tee $O/ab_cog_mfmasum.txt
SVG_ATTN_LIB=$L/libsvgattn_ms64.so timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_prescaled.py -q -k "64 or cog" -x 2>&1 | tail -3 | tee $O/pytest_d64_mfmasum.txt
SVG_ATTN_LIB=$L/libsvgattn_rs8.so timeout 300 python -m pytest tests/test_gpu_fp8.py -q 2>&1 | tail -3 | tee $O/pytest_fp8_rowsum.txt
tail -5 $O/err.txt
