#!/bin/bash
# The reference's complete variable-block grid (864 cases) on the code in the tree.  OMP_NUM_THREADS matters: 12 xdist workers x an
# unbounded torch thread pool on the host oversubscribes the oracle (r06q: 66 % of the grid in 2500 s against 317 s for all of it).
tag=${1:-r06r}; O=gpurun_out/$tag; mkdir -p $O
SVG_FULL_GRID=1 OMP_NUM_THREADS=8 timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k full_reference_grid -n 12 > $O/varblock_fullgrid.txt 2>&1
echo "fullgrid rc=$?" >> $O/varblock_fullgrid.txt; grep -v amdgpu.ids $O/varblock_fullgrid.txt | tail -4
