#!/bin/bash
# round 3, last kernel call: two-phase bodies (pre-scaled bf16 / f16 and fp8) without the per-tile register copies of the loop-invariant
# C operand (in-place reference rewrite + asm first step) and without the zeroing moves of the e4m3 conversions.  Bit-exactness A/B
# against the previous build first; the GPU suite only if that is clean.
O=gpurun_out/r03zh; mkdir -p $O
L=$PWD/sparse-videogen_amd/lib
timeout 400 python tools/ab_bitexact.py $L/libsvgattn_prev.so $L/libsvgattn.so > $O/ab_bitexact.txt 2>&1; rc=$?; echo "rc=$rc" >> $O/ab_bitexact.txt; grep -v amdgpu.ids $O/ab_bitexact.txt
if [ $rc -eq 0 ]; then
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.txt 2>&1; tail -1 $O/pytest_gpu.txt
fi
