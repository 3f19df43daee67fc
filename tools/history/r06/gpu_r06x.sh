#!/bin/bash
# flash-kmeans assignment with the workgroup's halves one phase apart (SVG_KMEANS_PP) against the single-phase form (lib/libsvgattn_nopp.so,
# -DSVG_KMEANS_NO_PP): labels first (oracle / golden / fuzz tests), then the same-box A/B: assignment alone over K, the SVG2 stage and init
tag=${1:-r06x}; O=gpurun_out/$tag; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize_svg2.py tests/test_gpu_triton_golden.py tests/test_gpu_fuzz.py tests/test_gpu_reference_calls.py -q -m gpu -x -k "kmeans or svg2 or sap or dynamic or varblock_pipeline" > $O/pytest_kmeans.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_kmeans.txt
grep -v amdgpu.ids $O/pytest_kmeans.txt | tail -5
L=$PWD/sparse-videogen_amd/lib
for r in 1 2; do
  for l in libsvgattn libsvgattn_nopp; do
    echo "== $l round $r"; SVG_ATTN_LIB=$L/$l.so timeout 200 python tools/kmeans_assign_probe.py 2>/dev/null | grep -E '"K": (320|1000|4096)|sclk' | tee -a $O/probe_$l.jsonl
  done
done
for r in 1 2 3; do
  for l in libsvgattn libsvgattn_nopp; do
    echo "== $l round $r"; timeout 100 tools/native_svg2 --lib $L/$l.so --geom wan720p --two-streams --warm 2 --reps 5 2>/dev/null | tee -a $O/svg2_$l.jsonl | cut -c1-700
  done
done
