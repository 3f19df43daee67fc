#!/bin/bash
# strided tensors on the head_dim-64 two-phase body as well: parity, then the CogVideoX A/B
tag=${1:-r06u}; O=gpurun_out/$tag; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_strided.py tests/test_gpu_kernels.py tests/test_gpu_processors.py tests/test_gpu_reference_calls.py tests/test_gpu_prescaled.py -q -m gpu -x > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_subset.txt
grep -v amdgpu.ids $O/pytest_subset.txt | tail -6
timeout 600 python tools/ab_strided.py cog 6 2> $O/ab.err | tee $O/ab_strided_cog.jsonl
timeout 600 python tools/ab_strided.py hy 6 2>> $O/ab.err | tee -a $O/ab_strided_cog.jsonl
tail -n 3 $O/ab.err
