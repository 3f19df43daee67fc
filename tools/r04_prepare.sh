#!/bin/bash
# Round 4, step 0 (on the build machine, no GPU needed): tagged A/B builds of the library for the experiments prepared at the end of
# round 3 — every switch is off in the product build and leaves its kernels' listings byte-identical (DESIGN.md §7 work list).
#   lib/libsvgattn_km2.so   -DSVG_KMEANS_V2=1        k-means assignment: operand ring four reads ahead, two-chain arg-max   (bit-identical results expected)
#   lib/libsvgattn_rs8.so   -DSVG_F8_MFMA_ROWSUM=1   fp8 gathering body (SVG2): row sums on the matrix pipe                 (changes the normaliser)
#   lib/libsvgattn_ms64.so  -DSVG_PP2_MFMASUM=1      16-bit two-phase body at head_dim 64 (CogVideoX): the same              (changes the normaliser)
# then:  gpurun --timeout 900 -- 'bash tools/r04_first_call.sh'   and   gpurun --timeout 900 -- 'bash tools/r04_second_call.sh'
set -e
cd "$(dirname "$0")/.."
SVG_EXTRA_HIPCC_FLAGS=-DSVG_KMEANS_V2=1 python sparse-videogen_amd/build.py --tag km2 | tail -1
SVG_EXTRA_HIPCC_FLAGS=-DSVG_F8_MFMA_ROWSUM=1 python sparse-videogen_amd/build.py --tag rs8 | tail -1
SVG_EXTRA_HIPCC_FLAGS=-DSVG_PP2_MFMASUM=1 python sparse-videogen_amd/build.py --tag ms64 | tail -1
python sparse-videogen_amd/build.py | tail -1
