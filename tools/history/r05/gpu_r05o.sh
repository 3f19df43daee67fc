#!/bin/bash
tag=${1:-r05o}; O=gpurun_out/$tag; mkdir -p $O
t0=$(date +%s)
timeout 1700 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$? $(( $(date +%s) - t0 )) s" >> $O/pytest_gpu.txt; tail -15 $O/pytest_gpu.txt
