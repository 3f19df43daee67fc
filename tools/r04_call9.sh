#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_m16.py tests/test_gpu_kernels.py -q -x -k "m16 or launch_order or device_switch" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/pytest_m16.txt
timeout 300 python tools/ab_m16.py 4 2 2>&1 | grep -v amdgpu.ids | tee $O/ab_m16.txt
