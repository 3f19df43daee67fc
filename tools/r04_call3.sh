#!/bin/bash
# Round 4, GPU call 3: energy table with the 16x16x32 MFMA shape in the full mix; cyclic start with stagger (time + PMC traffic per setting)
O=gpurun_out/r04c; mkdir -p $O
timeout 200 tools/energy_table 77 40000 2>&1 | tee $O/energy_table.txt
SVG_AB_ROTS=0,2,3,1 timeout 400 python tools/ab_rotate.py 4 2>&1 | grep -v amdgpu.ids | tee $O/ab_rotate.txt
bash tools/pmc_rotate.sh $O/pmc plain "0 1 2 3" 2>&1 | tail -6 | tee $O/pmc_rotate.txt
